/* ekaid_hip.h -- C ABI of libekaid_hip.so: the MI355X (gfx950) kernels behind the EKAID
 * difference-VQA forward/backward path.
 *
 * The reference (Holipori/EKAID) is pure Python/PyTorch and has NO FFI of its own: every entry
 * point below replaces a run of ATen ops inside one reference function, cited as
 * "ref: <file>:<lines>" relative to /root/reference/model.  The binding a maintainer of the
 * reference would add is a ctypes stub; ekaid_amd/hip.py is that stub.
 *
 * Conventions
 *   - plain C types only: device pointers + sizes, no torch/C++ types;
 *   - every tensor is contiguous row-major float32 on the current device unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     enqueues work on that stream -- no allocation, no host synchronisation, so calls are
 *     capturable into a hipGraph;
 *   - outputs and workspaces are caller-allocated;
 *   - return 0 on success, negative on error (ekaid_last_error() holds the message);
 *   - dropout masks are a pure function of (seed, element index): forward and backward pass the
 *     same seed, nothing is stored.  p == 0 disables dropout.
 */
#ifndef EKAID_HIP_H
#define EKAID_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKAID_ABI_VERSION 9

int ekaid_version(void);
const char* ekaid_last_error(void);        /* thread-local */

/* Per-device state.  The library keeps exactly one small state record per HIP device (GEMM arithmetic selection,
 * dropout epoch pointer, "kernel attribute already raised" flags), every field atomic: calls from different host
 * threads or on different streams do not race.  ekaid_init(device) creates the record eagerly (it is also created on
 * the first call made with that device current); all other entry points act on the calling thread's CURRENT device.
 * The reference has no counterpart (single process, single GPU: train_mimic.py:2). */
int ekaid_init(int device_id);

/* Dropout under hipGraph replay.  Every dropout site takes its 64-bit seed BY VALUE, which a captured graph freezes.
 * After ekaid_set_dropout_epoch_ptr(p) (p: one 64-bit word in device memory, NULL to switch off) every site of the
 * current device uses  seed + p[0] * 0x9E3779B97F4A7C15  instead; ekaid_dropout_epoch_advance enqueues p[0] += by on a
 * stream (put it at the head of the captured step): each replay draws fresh masks, forward and backward of one replay
 * agree.  ref: the reference draws fresh masks every iteration through torch's global RNG (nn.Dropout, F.dropout). */
int ekaid_set_dropout_epoch_ptr(const unsigned long long* epoch_dev);
int ekaid_dropout_epoch_advance(unsigned long long* epoch_dev, unsigned long long by, void* stream);

/* activation codes for fused epilogues */
#define EKAID_ACT_NONE 0
#define EKAID_ACT_RELU 1
#define EKAID_ACT_TANH 2
#define EKAID_ACT_SIGMOID 3

/* ------------------------------------------------------------------------------------------
 * Dense contraction on the fp32 matrix cores.
 *   C = dropout( act( alpha * alpha_dev[0] * op(A).op(B)  [+ C]  [+ bias[n]]  [+ rowmask[m] * R[m / rdiv][n]] ) )  [+ post[m][n]]
 * transA=0: A is [M][K]; transA=1: A is [K][M].  transB=1: B is [N][K]; transB=0: B is [K][N].
 * Two batch dimensions (batch1 x batch2) with element strides; batch index z -> (z / batch2, z % batch2).
 * ref: every nn.Linear / FCNet / matmul on the path -- modules.py:195-196,278-303,
 *      graph_att.py:80,92, graph_att_layer.py:79,90,105,167,174, language_model.py:113,138-142,
 *      dynamic_speaker_change_pos.py:99-125,238; their autograd backward (dgrad/wgrad).
 * ------------------------------------------------------------------------------------------ */
typedef struct ekaid_gemm {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int transA, transB;
    int batch1, batch2;
    long long strideA1, strideA2, strideB1, strideB2, strideC1, strideC2;
    float alpha;
    const float* alpha_dev; /* optional device scalar multiplied into alpha (weight-norm g/||v||) */
    int accumulate;         /* 1: C += ... (before bias/residual/act) */
    const float* bias;      /* [N] or NULL */
    const float* R;         /* optional residual, row (m / rdiv), leading dim ldr */
    int ldr;
    int rdiv;
    const float* rowmask;   /* optional [M] multiplier of the residual row */
    int act;
    float drop_p;           /* > 0: dropout of the activated output, keep(seed, m * N + n) / (1 - p) */
    uint64_t drop_seed;
    float* a_rowsum;        /* optional, needs transA: a_rowsum[m] += sum_k op(A)[m][k] (f32, fixed order).  The bias gradient of a
                               linear layer rides on its weight-gradient product dW = dY^T X this way (A = dY): dY is not read a
                               second time by a column-sum kernel.  ekaid_gemm_f32 only; counted by ekaid_gemm_workspace_bytes. */
    const float* post;      /* optional (one batch): C = dropout(act(...)) + post[m * ldpost + n] -- a gradient of the same tensor that joins
                               this input-gradient product behind its dropout, instead of an n-ary sum kernel afterwards */
    int ldpost;
    const float* a_amax;    /* optional DEVICE vectors (ABI 9): a_amax[m] >= max_k |op(A)[m][k]| (M entries), b_amax[n] >= max_k |op(B)[k][n]| */
    const float* b_amax;    /* (N entries); ekaid_amax_rc computes them.  Under EKAID_GEMM_F16X2 an unbatched product that carries both runs
                               as the two-term f16 split (3 matrix products per product); one that does not keeps the exact 3-way bf16
                               split.  Ignored in the other modes.  A bound that is too SMALL overflows f16 (Inf / NaN in that row / column);
                               one that is too large by 2^k costs k bits of the 2^17 : 1 range inside a row / column (see below). */
} ekaid_gemm_t;

size_t ekaid_gemm_workspace_bytes(const ekaid_gemm_t* g);
int ekaid_gemm_f32(const ekaid_gemm_t* g, void* workspace, size_t workspace_bytes, void* stream);
/* TWO independent products (neither reads what the other writes) in ONE launch: workgroups [0, tiles of g0) work on g0, the rest on
 * g1, every product keeps the tiling, K split, epilogue and summation order of its own plan -- results are BIT-IDENTICAL to
 * ekaid_gemm_f32(g0); ekaid_gemm_f32(g1).  Why: the detector's products are cut into 1152 tiles of 128x128 for 768 resident workgroup
 * slots (1.5 waves); two of them in one grid are three whole waves, and a launch's fixed cost (~22 us) is paid once.  Taken for the
 * query / key projections of a graph-attention layer, the context / gate pairs of the fusion tail, and the two input- and
 * weight-gradient products of a two-input layer.  Pairs outside the pair kernel's envelope (different layouts or tiles, batched or
 * tall-skinny products) run as two launches.  workspace_bytes >= ekaid_gemm_pair_workspace_bytes(g0, g1).
 * ref: graph_att_layer.py:79,90 (query / key), modules.py:278-286 (context / gate), graph_att.py:80 (self_weights over [v | q]). */
size_t ekaid_gemm_pair_workspace_bytes(const ekaid_gemm_t* g0, const ekaid_gemm_t* g1);
int ekaid_gemm_pair_f32(const ekaid_gemm_t* g0, const ekaid_gemm_t* g1, void* workspace, size_t workspace_bytes, void* stream);
/* What the planner launches for g (host-side, no launch): *tile = 1 (64x64), 2 (128x128), 3 (128x64), 4 (64x128) or 0 (tall-skinny
 * reduction kernels), *splitk = K slabs.  For profiles: bench.py attributes every timed launch to the kernel instantiation that ran. */
int ekaid_gemm_plan_query(const ekaid_gemm_t* g, int* tile, int* splitk);

/* ------------------------------------------------------------------------------------------
 * Raw partial products for consumers that finish the reduction themselves ("slabs").
 * The small recurrent GEMMs (M = batch rows) are cut along K so that their grids fill the chip; instead of a reduce kernel
 * between the GEMM and its consumer, the consumer (an LSTM / GRU cell, a row-local chain, a bias + activation epilogue) sums
 * the slabs itself, in a fixed order (deterministic).
 *   ekaid_gemm_slab_count(g)  number of slabs n the planner cuts g into (>= 1; host-side, no launch);
 *   ekaid_gemm_slabs(g, ...)  slabs[s][M][N] (s < n, row-major, leading dim N) with sum_s slabs[s] = op(A).op(B): no alpha,
 *                             bias, residual, activation or dropout (those fields of g are ignored), g->C unused;
 *                             slab_bytes >= n * M * N * 4.  One batch only.
 *   ekaid_gemm_slabs2         two independent products in ONE launch when both use the 64x64-tile kernel and the same operand
 *                             layout (otherwise two launches, same results).
 * A consumer addresses a slab set through ekaid_slabs_t: element (row, col) of slab s is p[s * stride + row * ld + col];
 * `p` may point at a column block of the slabs, n = 0 (or p = NULL) means "no such source"; a plain matrix is n = 1.
 * ref: the nn.Linear / nn.LSTMCell / nn.GRU products inside the time loops, dynamic_speaker_change_pos.py:94-131,
 *      language_model.py:106-115.
 * ------------------------------------------------------------------------------------------ */
typedef struct ekaid_slabs {
    const float* p;
    int n;
    long long stride;       /* elements between consecutive slabs (M * N of the producing GEMM) */
    int ld;                 /* leading dimension inside a slab (N of the producing GEMM) */
} ekaid_slabs_t;
int ekaid_gemm_slab_count(const ekaid_gemm_t* g);
int ekaid_gemm_slabs(const ekaid_gemm_t* g, float* slabs, size_t slab_bytes, void* stream);
int ekaid_gemm_slabs2(const ekaid_gemm_t* g0, float* slabs0, size_t slab_bytes0, const ekaid_gemm_t* g1, float* slabs1,
                      size_t slab_bytes1, void* stream);

/* Arithmetic of ekaid_gemm_f32 (per device; both take and return f32 and accumulate in f32):
 *   EKAID_GEMM_F32_MFMA   (0)  v_mfma_f32_32x32x2_f32, an exact f32 fma chain;
 *   EKAID_GEMM_BF16_SPLIT (1)  every f32 operand element is split EXACTLY into three bf16 values (8+8+8 significant
 *                              bits) and the six products above 2^-24 relative are formed on v_mfma_f32_32x32x16_bf16
 *                              with f32 accumulation: f32-level accuracy (dropped terms < 2^-23 |a.b| per product) at
 *                              2.67x the matrix rate.  Default; the environment variable EKAID_GEMM_MODE=f32 selects 0.
 * Inputs must be finite in mode 1: an Inf operand gives NaN where mode 0 gives Inf (Inf - Inf in the split).  Unreachable on
 * this path -- every GEMM operand is a finite activation, weight or gradient; a non-finite one is already a failed run, and
 * EKAID_CHECK_FINITE=1 (ekaid_amd/ops.py) asserts exactly that per module output.
 *   EKAID_GEMM_F16X2      (2)  products whose descriptor carries a_amax AND b_amax: every operand element, scaled by an exact
 *                              power of two s that brings the largest magnitude of its row of op(A) / column of op(B) into
 *                              [2^14, 2^15) (the scales leave the result through the epilogue, exactly), is split into TWO f16
 *                              values  v s = h + l + e,  h = f16(v s), l = f16(v s - h)  (round to nearest: 11 + 11 significant
 *                              bits, |e| <= 2^-22 |v s|), and the three products h h + h l + l h are formed on
 *                              v_mfma_f32_32x32x16_f16 with f32 accumulation; the dropped l l is below 2^-22 |a b|.  Half the
 *                              matrix work of mode 1.  Per-product error <= ~2^-21 |a b| worst case, ~2^-25 rms -- below the
 *                              accumulation error of an f32 fma chain from K ~ 64 on (measured against float64:
 *                              tests/test_hip_ops.py::test_gemm_split_accuracy).  Elements more than 2^17 below the largest of
 *                              their row / column keep an absolute accuracy of 2^-40 of that largest (f16 subnormals are not
 *                              flushed by the matrix pipe).  Products WITHOUT bounds run as mode 1, so mode 2 is mode 1 plus a fast path for the
 *                              call sites that supply bounds.  EKAID_GEMM_MODE=f16x2 selects it. */
#define EKAID_GEMM_F32_MFMA 0
#define EKAID_GEMM_BF16_SPLIT 1
#define EKAID_GEMM_F16X2 2
/* Row and column maxima of |x|: rowmax[r] = max_c |x[r * ld + c]| (r < rows), colmax[c] = max_r |x[r * ld + c]| (c < cols); either
 * output may be NULL.  These are the operand bounds of EKAID_GEMM_F16X2: for a product, a_amax = the row maxima of A when transA = 0
 * (A stored [M][K]) or its column maxima when transA = 1 (stored [K][M]); b_amax = the row maxima of B when transB = 1 (stored [N][K])
 * or its column maxima when transB = 0 (stored [K][N]); bounds taken over a larger block of the same rows / columns stay valid.
 * Exact and order-independent (integer max of the magnitude's bit pattern); an all-zero row gives 0, a NaN element NaN.  One launch
 * for up to 32 matrices (x, rows, cols, ld, rowmax, colmax: HOST arrays).  The outputs are atomic-max targets -- every column output, and
 * the row output of a matrix wider than 256 columns (narrower ones are stored) -- so they must start at zero: outputs_zeroed = 0 clears
 * them here (one memset node each), outputs_zeroed = 1 says the caller hands in memory that already holds zeros (slices of a pool cleared
 * once per step: a bounds pass is then ONE graph node -- at 32 pairs per GPU the step is a chain of ~500 small nodes and every one
 * counts).  cols, ld multiples of 4 and 16-byte aligned x, else EKAID_ERR_UNSUPPORTED (the caller then leaves the product on the exact
 * split).  No reference counterpart. */
int ekaid_amax_rc_multi(const float* const* x, const long long* rows, const int* cols, const int* ld, float* const* rowmax,
                        float* const* colmax, int count, int outputs_zeroed, void* stream);
int ekaid_amax_rc(const float* x, long long rows, int cols, int ld, float* rowmax, float* colmax, int outputs_zeroed, void* stream);
int ekaid_set_gemm_mode(int mode);
int ekaid_get_gemm_mode(void);

/* ------------------------------------------------------------------------------------------
 * Weight normalisation, dim=None:  W = v * g / ||v||_F.   ref: fc.py:26-33 (weight_norm(Linear, dim=None))
 * fwd: scale_out[0] = g / ||v||,  norm_out[0] = ||v||.
 * bwd: given dscale (gradient of the loss w.r.t. scale): dg[0] = dscale / ||v||,
 *      dv (+)= -dscale * g / ||v||^3 * v      (accumulate_dv: 1 adds into dv, 0 overwrites)
 * scale_dot: dot_out[0] = <G, w>; then G *= scale_dev[0]   (G = unscaled wgrad; gives dW and dscale)
 * ------------------------------------------------------------------------------------------ */
int ekaid_wn_scale_fwd(const float* v, const float* g, long long n, float* scale_out, float* norm_out, void* stream);
int ekaid_wn_scale_bwd(const float* v, const float* g, const float* norm, const float* dscale, long long n,
                       float* dv, int accumulate_dv, float* dg, void* stream);
int ekaid_scale_dot(float* G, const float* w, const float* scale_dev, long long n, float* dot_out, void* stream);
/* The same for a whole model at once and fused:
 * wn_scale_multi: scales[i] = g_i / ||v_i||, norms[i] = ||v_i|| for `count` <= 32 layers in two launches (v, g, n are HOST
 *                 arrays of device pointers / element counts) -- the weights do not change inside a training step, so all
 *                 norms are taken once at its start instead of two launches per layer.
 * wn_grad_fused:  backward of one layer in two launches: G holds the UNSCALED weight gradient dy^T x on entry and
 *                 dv = scale * G - (<G, v> g / ||v||^3) v on exit; dg[0] = <G, v> / ||v||   (= scale_dot + wn_scale_bwd). */
int ekaid_wn_scale_multi(const float* const* v, const float* const* g, const long long* n, int count, float* scales, float* norms,
                         void* stream);
int ekaid_wn_grad_fused(float* G, const float* v, const float* g, const float* norm, const float* scale, long long n, float* dg,
                        void* stream);
/* wn_grad_multi:  wn_grad_fused for `count` <= 32 layers in two launches (HOST arrays of device pointers / element counts): the raw
 *                 weight gradients of every weight-normed layer of a step are turned into dv / dg together at the end of backward
 *                 (the transform is linear in the raw gradient, so it commutes with the data-parallel all-reduce). */
int ekaid_wn_grad_multi(float* const* G, const float* const* v, const float* const* g, const float* const* norm, const float* const* scale,
                        const long long* n, float* const* dg, int count, void* stream);

/* column sum of an [M][N] matrix (bias gradient), out[N]; row-block partials summed in a fixed order (deterministic) */
int ekaid_colsum(const float* x, int M, int N, int ldx, float* out, void* stream);
/* out[N] += column sums (the bias gradient accumulated straight into its zeroed slot of the flat gradient buffer, or on
 * top of an earlier contribution of a shared bias); same fixed-order partial sums */
int ekaid_colsum_acc(const float* x, int M, int N, int ldx, float* out, void* stream);
/* out = xs[0] + ... + xs[k-1] (1 <= k <= 8 device arrays of n floats, xs itself is a HOST array; out may alias xs[0]):
 * the gradient of a tensor with several consumers in one pass (ref: autograd's pairwise accumulation of fan-out gradients,
 * e.g. the encoder output feeding context/gate/attention/pooling, modules.py:266-303) */
int ekaid_sum_n(const float* const* xs, int k, float* out, long long n, void* stream);
/* `steps` dropout masks in one launch: y[t][i] = x[t * x_step + i] * keep(seed + t, i) / (1 - p), i < n  (x_step = 0: one
 * input under a fresh mask per decoding step, dynamic_speaker_change_pos.py:101; x_step = n: its backward) */
int ekaid_dropout_steps(const float* x, float* y, long long n, int steps, long long x_step, float p, uint64_t seed, void* stream);
/* dz = dy * act'(y) with the derivative written through the activation output y */
int ekaid_act_bwd(const float* dy, const float* y, float* dz, long long n, int act, void* stream);
/* y = x * keep(seed, i) / (1 - p)      (also its own backward)   ref: every nn.Dropout on the path */
int ekaid_dropout(const float* x, float* y, long long n, float p, uint64_t seed, void* stream);
/* out = a*x + b*y (y may be NULL) */
int ekaid_axpby(const float* x, const float* y, float* out, long long n, float a, float b, void* stream);
/* out = x * y */
int ekaid_mul(const float* x, const float* y, float* out, long long n, void* stream);
/* rowsum over the last dim: out[r] = sum_c x[r][c];  mask variant: out[r] = (sum != 0)
 * ref: graph_att.py:89 (condensed adjacency), relation_encoder.py:24-27 (zero-row test) */
int ekaid_rowsum(const float* x, long long rows, int cols, float* out, int as_nonzero_mask, void* stream);
/* mask[r] = (sum of row r != 0) exactly as ekaid_rowsum(..., 1) forms it, and y = dropout(x) (mask index = flat index), one pass:
 * the zero-row mask of q_expand_v_cat (relation_encoder.py:19-29) and the input dropout of the FCNet that reads x (fc.py:36-37). */
int ekaid_rowmask_dropout(const float* x, long long rows, int cols, float* mask, float* y, float p, uint64_t seed, void* stream);
/* out = (a*x + b*x) + c*x, evaluated in that order (the reference's three "parallel" encoder outputs are one tensor: modules.py:234-235). */
int ekaid_self_mix3(const float* x, float* out, long long n, float a, float b, float c, void* stream);
/* out[r][c] = mask[r] * q[(r / group) % qrows][c]     ref: relation_encoder.py:19-29 (q_expand_v_cat, q half;
 * qrows = batch size lets the main and the reference image share one question matrix)
 * bwd: dq[j][c] = sum over rows r with (r / group) % qrows == j of mask[r] * dout[r][c] */
int ekaid_expand_rows_fwd(const float* q, const float* mask, long long rows, int group, int qrows, int cols, float* out,
                          float p /* fused dropout of the expanded rows, 0 = off */, uint64_t seed, void* stream);
int ekaid_expand_rows_bwd(const float* dout, const float* mask, long long rows, int group, int qrows, int cols, float* dq,
                          float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------------
 * Graph attention core.   ref: graph_att_layer.py:105-172 (+ graph_att.py:76,88 for `dir`)
 *   q,S : [R][N][A], k : [R][M][A], cond/lbias : [R][N][N] (condensed adjacency and
 *   label bias in the ORIGINAL orientation; dir=1 reads them transposed), geo : [R][N][M][H] relu'd
 *   pair_pos_fc1 output or NULL.  P : [R][N][H][M] softmax (returned by the layer),
 *   outT : [R][N][H*A] = per-head P.S (input of linear_out_2).
 * masked_softmax_{fwd,bwd}: the edge-softmax alone, on precomputed scaled affinities aff [R][N][H][M].
 *   bwd writes daff, dgeo (may be NULL) and ACCUMULATES dlbias[R][N][N] (caller zero-fills).
 * ------------------------------------------------------------------------------------------ */
int ekaid_gat_fused_fwd(const float* q, const float* k, const float* S, const float* cond, const float* lbias,
                        const float* geo, int R, int N, int M, int A, int H, int dir, float* P, float* outT,
                        void* stream);
/* dq[r][n][h,:] = scale * sum_m daff[r][n][h][m] k[r][m][h,:],  dk[r][m][h,:] = scale * sum_n daff[r][n][h][m] q[r][n][h,:]
 * (gradients of the scaled affinities q_h . k_h^T; daff from ekaid_masked_softmax_bwd) in one launch.  ERR_UNSUPPORTED outside
 * N <= 40, M <= N, A % (4 H) == 0, (N + M) * H * 160 bytes of LDS <= 64 KB: the caller composes them from two batched
 * ekaid_gemm_f32 calls then (faster there).
 * ref: autograd of graph_att_layer.py:79-108 */
int ekaid_gat_dqk(const float* daff, const float* q, const float* k, int R, int N, int M, int A, int H, float scale, float* dq,
                  float* dk, void* stream);
/* Backward of the aggregation and of the edge softmax in one launch, one workgroup per slice (graph_att_layer.py:146-172, autograd):
 *   dP[r][n][h][m] = <doutT[r][n][h][:], S[r][m][:]> (+ dP_ext, the gradient arriving on the returned softmax, or NULL)
 *   dS[r][m][:]    = sum_{n,h} P[r][n][h][m] doutT[r][n][h][:]            (rows m < M; the caller zeroes rows M .. N-1 when M < N)
 *   daff, dlbias, dgeo as ekaid_masked_softmax_bwd computes them from P and dP.
 * doutT (R, N, H, A) is read once; exact f32 products (v_mfma_f32_16x16x4_f32).  geo and dgeo both NULL or both given.
 * ERR_UNSUPPORTED outside M <= N, M <= 48, N H <= 144 rows, A % 32 == 0: the caller composes the same from two batched
 * ekaid_gemm_f32 calls + ekaid_masked_softmax_bwd then (also the faster form below ~256 slices: one workgroup per slice). */
int ekaid_gat_bwd_core(const float* doutT, const float* S, const float* P, const float* dP_ext, const float* cond, const float* geo,
                       int R, int N, int M, int A, int H, int dir, float* dS, float* daff, float* dlbias, float* dgeo, void* stream);
int ekaid_masked_softmax_fwd(const float* aff, const float* cond, const float* lbias, const float* geo, int R,
                             int N, int M, int H, int dir, float* P, void* stream);
int ekaid_masked_softmax_bwd(const float* P, const float* dP, const float* cond, const float* geo, int R, int N,
                             int M, int H, int dir, float* daff, float* dlbias, float* dgeo, void* stream);

/* float64 box geometry -> float32 sinusoidal pair embedding [R][N][M][feat_dim]
 * ref: utils/mimic_utils.py:152-208, modules.py:162-166.  bb is a DEVICE pointer to [R][N][4] doubles;
 * dim_mat a device array of feat_dim/8 floats = 1000^(8t/feat_dim) (the reference's float32 table). */
int ekaid_geom_embed(const double* bb, const float* dim_mat, int R, int N, int M, int feat_dim, float* out, void* stream);
/* The same with the input dropout of the embedding's reader (pair_pos_fc1: graph_att_layer.py:121, fc.py:36-37) applied as the
 * embedding is written: out *= keep(seed, flat index) / (1 - p).  The embedding has no gradient, so nothing else changes. */
int ekaid_geom_embed_dropout(const double* bb, const float* dim_mat, int R, int N, int M, int feat_dim, float* out, float p,
                             uint64_t seed, void* stream);

/* v_new = v + relu(dropout(2*o))       ref: graph_att.py:98-104 (quirk Q-B) + relation_encoder.py:80-81,128-129 (Q-A)
 * bwd: do = dvnew * 2 * keep/(1-p) * [o*keep > 0]   (dv = dvnew) */
int ekaid_residual_relu2_fwd(const float* v, const float* o, float* vnew, long long n, float p, uint64_t seed, void* stream);
int ekaid_residual_relu2_bwd(const float* o, const float* dvnew, float* d_o, long long n, float p, uint64_t seed, void* stream);

/* out = dropout(sigmoid(g2 + g1)) * dropout(tanh(c2 + c1))      ref: modules.py:278-288
 * c2/g2: n elements (context2/gate2 of both images, bias included); c1/g1: n1 elements (context1/gate1 of the
 * difference, shared by both images and computed once), broadcast: index i -> i % n1. */
int ekaid_gate_fuse_fwd(const float* c2, const float* g2, const float* c1, const float* g1, float* out, long long n,
                        long long n1, float p, uint64_t seed_c, uint64_t seed_g, void* stream);
int ekaid_gate_fuse_bwd(const float* c2, const float* g2, const float* c1, const float* g1, const float* dout, float* dc2,
                        float* dg2, float* dc1, float* dg1, long long n, long long n1, float p, uint64_t seed_c,
                        uint64_t seed_g, void* stream);

/* out[g][c] = sum_i w[g][i] * x[g][i][c]      ref: modules.py:305-308 (region pooling), language_model.py:151 (bmm) */
int ekaid_weighted_pool_fwd(const float* w, const float* x, int G, int n, int C, float* out, void* stream);
int ekaid_weighted_pool_bwd(const float* w, const float* x, const float* dout, int G, int n, int C, float* dw,
                            float* dx, void* stream);

/* Question self-attention weights with the reference's batch-coupled softmax (quirk Q-C):
 * S = softmax over the batch axis of atten^T (L x B); weight = S's contiguous memory re-read as (B, L).
 * ref: language_model.py:149.  S_out [L][B] is saved for the backward. */
int ekaid_qatt_weights_fwd(const float* atten, int B, int L, float* S_out, float* weight, void* stream);
int ekaid_qatt_weights_bwd(const float* S, const float* dweight, int B, int L, float* datten, void* stream);

/* two-table embedding lookup / single table.  ref: language_model.py:48-53, dynamic_speaker_change_pos.py:153-156,229
 * fwd: out[t][col_off + e] = act(table[tok[t]][e]); bwd: dtable[tok[t]][e] += dout[t][col_off+e] * act'
 *      (rows equal to padding_idx receive no gradient) */
int ekaid_embedding_fwd(const int64_t* tok, long long ntok, const float* table, int E, float* out, int ld_out,
                        int col_off, int relu, void* stream);
/* bwd takes the number of table rows: one wave per (row, 64 columns) adds the matching positions in position order
 * (deterministic, no float atomics; serial in the number of positions per row -- fine for evenly used ids).  The layers use
 * the GEMM form instead, whose cost does not depend on the id distribution (padding / start ids hit thousands of
 * positions):  onehot [ntok][vocab] (ekaid_onehot, padding rows zero);  raw [vocab][E] = onehot^T . dout (ekaid_gemm_f32,
 * transA = 1: split-K with a fixed-order reduction);  dtable += raw * (table > 0 if relu) (ekaid_masked_add). */
int ekaid_onehot(const int64_t* tok, long long ntok, int vocab, long long padding_idx, float* out, void* stream);
int ekaid_masked_add(const float* raw, const float* table, int relu, float* dtable, long long n, void* stream);
int ekaid_embedding_bwd(const int64_t* tok, long long ntok, const float* table, int vocab, const float* dout, int E, int ld_out,
                        int col_off, int relu, long long padding_idx, float* dtable, void* stream);

/* GRU / LSTM cell point-wise parts (the gate GEMMs go through ekaid_gemm_f32).
 * ref: language_model.py:66-70,113 (nn.GRU, gate order r,z,n), dynamic_speaker_change_pos.py:57,88,102,126 (nn.LSTMCell, i,f,g,o)
 * gi/gh/gx may be strided views (leading dims ld*). save: GRU [B][4H] = r,z,n,gh_n ; LSTM [B][4H] = i,f,g,o.
 * gru bwd: upstream gradient = dhnew (strided, may be NULL) + dh_carry (contiguous [B][H], may be NULL);
 * dh receives only the direct (z * d) term -- the caller adds dgh . W_hh with an accumulating GEMM. */
int ekaid_gru_cell_fwd(const float* gi, int ldgi, const float* gh, int ldgh, const float* h, int ldh, int B, int H,
                       float* hnew, int ldhn, float* save, void* stream);
int ekaid_gru_cell_bwd(const float* dhnew, int lddh, const float* dh_carry, const float* save, const float* h, int ldh,
                       int B, int H, float* dgi, int lddgi, float* dgh, float* dh, void* stream);
int ekaid_lstm_cell_fwd(const float* gx, const float* gh, int ldgh, const float* c, int B, int H, float* h2, float* c2,
                        float* save, void* stream);
/* upstream gradient on h2 = dh2 + dh2b (either may be NULL); dgates written with leading dim lddg */
int ekaid_lstm_cell_bwd(const float* dh2, const float* dh2b, const float* dc2, const float* save, const float* c,
                        const float* c2, int B, int H, float* dgates, int lddg, float* dc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused row-local chains of one decoder step (one workgroup per batch row).
 * ref: DynamicCore.forward, dynamic_speaker_change_pos.py:104-125 and its autograd backward.
 *
 * spk_modmix_fwd : mw = softmax(h . w_fc^T + b_fc) over the 3 modules (:104-106); att = mw0*bef + mw1*diff + mw2*aft
 *                  (:108-113), written to `att` (contiguous, B x A) and, if att2 != NULL, to att2 (leading dim ld2).
 * spk_modmix_bwd : d_att = d_att_a[b*lda + k] (+ d_gated*gate if d_gated != NULL: the product rule of gated = gate*att);
 *                  d_bef/d_diff/d_aft += mw_c * d_att; dz_out = softmax backward of <d_att, feat_c>; dh += dz . w_fc.
 * spk_pos_fwd    : v = dropout(relu(x + b1), p_v) (mask index b*R + k); dpos = dropout(w_wp v + b_wp, p_d) (mask index
 *                  b*P + j); pw = softmax(dpos); pp = w_p2 pw + b_p2 (:115-121).  Saves vp (= v), dp (= dpos), pw; pp is
 *                  written with leading dim ldpp.  R <= 2048, P <= 64.
 * spk_pos_bwd    : inverse chain: dpp (leading dim lddpp) -> dwp_out (B x P, gradient at the weight_pos output before
 *                  its dropout's mask was applied in forward, i.e. after dropout backward) and dx_out (gradient at the
 *                  pos1 pre-activation, leading dim lddx); dhead (B x P, may be NULL) is added at dpos.
 * spk_gate_bwd   : dz = d_gated * att * gate * (1 - gate)     (gated = sigmoid(z) * att, :122-125)
 * ------------------------------------------------------------------------------------------ */
int ekaid_spk_modmix_fwd(const float* h, int B, int R, const float* w_fc, const float* b_fc, const float* bef, const float* diff,
                         const float* aft, int A, float* mw, float* att, float* att2, int ld2, void* stream);
int ekaid_spk_modmix_bwd(const float* mw, const float* bef, const float* diff, const float* aft, int B, int A,
                         const float* d_att_a, int lda, const float* d_gated, const float* gate, const float* w_fc, int R,
                         float* dz_out, float* d_bef, float* d_diff, float* d_aft, float* dh, void* stream);
int ekaid_spk_pos_fwd(const float* x, int ldx, const float* b1, int B, int R, const float* w_wp, const float* b_wp, int P,
                      const float* w_p2, const float* b_p2, float p_v, uint64_t seed_v, float p_d, uint64_t seed_d,
                      float* vp, float* dp, float* pw, float* pp, int ldpp, void* stream);
int ekaid_spk_pos_bwd(const float* dpp, int lddpp, const float* pw, const float* dhead, int B, int P, const float* w_p2,
                      const float* w_wp, const float* vp, int R, float p_v, uint64_t seed_v, float p_d, uint64_t seed_d,
                      float* dwp_out, float* dx_out, int lddx, void* stream);
int ekaid_spk_gate_bwd(const float* d_gated, const float* att, const float* gate, long long n, float* dz, void* stream);

/* ------------------------------------------------------------------------------------------
 * Slab-consuming forms of the recurrent steps (csrc/steps.hip): every `*_src` / `carry` argument is a slab set produced by
 * ekaid_gemm_slabs; sources are summed (all slabs of all sources, fixed order) before the arithmetic of the plain form.
 *
 * slab_epilogue        out[r*ldo + c] = dropout(act(sum + bias[c]))      (mask index r*cols + c; up to 3 sources)
 *                      ref: the gate MLP's first layer, dynamic_speaker_change_pos.py:122-124 (Linear + ReLU + Dropout)
 * drop_act_slabs_bwd   out[r*ldo + c] = sum[r][c] * keep(seed, r*cols + c)/(1-p) * act'(y[r*ldy + c])
 * lstm_cell_slabs_fwd  gates = gx + sum + bias (i,f,g,o), then as ekaid_lstm_cell_fwd     (nn.LSTMCell, :57,88,102,126)
 * lstm_cell_slabs_bwd  upstream gradient on h2 = sum(carry) + dh_b (either may be absent), then as ekaid_lstm_cell_bwd
 * gru_cell_slabs_fwd   gh = sum + b_hh, then as ekaid_gru_cell_fwd                          (nn.GRU, language_model.py:113)
 * gru_cell_slabs_bwd   upstream = dhnew + dh_direct + sum(carry); writes dgi, dgh and dh_out = z * upstream (the direct term
 *                      the PREVIOUS step adds to its upstream gradient; the W_hh term arrives as that step's carry slabs)
 * gate_mul_slabs_fwd   gate = sigmoid(sum + bias); gated = gate * att                       (:122-125)
 * gate_slabs_bwd       d_gated = sum; dz = d_gated * att * gate * (1 - gate)
 * ------------------------------------------------------------------------------------------ */
int ekaid_slab_epilogue(const ekaid_slabs_t* src, int nsrc, const float* bias, int rows, int cols, int act, float p, uint64_t seed,
                        float* out, int ldo, void* stream);
int ekaid_drop_act_slabs_bwd(const ekaid_slabs_t* src, const float* y, int ldy, int rows, int cols, int act, float p, uint64_t seed,
                             float* out, int ldo, void* stream);
int ekaid_lstm_cell_slabs_fwd(const float* gx, const ekaid_slabs_t* src, int nsrc, const float* bias, const float* c, int B, int H,
                              float* h2, float* c2, float* save, void* stream);
int ekaid_lstm_cell_slabs_bwd(const ekaid_slabs_t* carry, const float* dh_b, const float* dc2, const float* save, const float* c,
                              const float* c2, int B, int H, float* dgates, int lddg, float* dc, void* stream);
int ekaid_gru_cell_slabs_fwd(const float* gi, int ldgi, const ekaid_slabs_t* gh, const float* b_hh, const float* h, int ldh, int B,
                             int H, float* hnew, int ldhn, float* save, void* stream);
int ekaid_gru_cell_slabs_bwd(const float* dhnew, int lddh, const float* dh_direct, const ekaid_slabs_t* carry, const float* save,
                             const float* h, int ldh, int B, int H, float* dgi, int lddgi, float* dgh, float* dh_out, void* stream);
int ekaid_gate_mul_slabs_fwd(const ekaid_slabs_t* src, const float* bias, const float* att, int B, int A, float* gate, float* gated,
                             void* stream);
int ekaid_gate_slabs_bwd(const ekaid_slabs_t* src, const float* att, const float* gate, int B, int A, float* dz, float* d_gated,
                         void* stream);

/* Row-local first half of ONE decoder step for the whole batch, one workgroup per batch row (descriptor by pointer: too
 * many operands for a flat signature).  ref: DynamicCore.forward, dynamic_speaker_change_pos.py:102-121.
 *   1. module-attention LSTM cell: gates = gxa + sum(gates_src) + b_hh  ->  h_out, c_out, gate_save (i,f,g,o)       (:102)
 *   2. mw = softmax(h_out . w_fc^T + b_fc); att = mw0*bef + mw1*diff + mw2*aft -> att (B x A) and att2 (ld2, may be NULL) (:104-113)
 *   3. v = dropout(relu(sum(pos_src) + b_p1), p_v); dpos = dropout(w_wp v + b_wp, p_d); pw = softmax(dpos);
 *      pp = w_p2 pw + b_p2 -> vp, dp, pw, pp (leading dim ldpp)                                                     (:115-121)
 * Dropout mask indices as ekaid_spk_pos_fwd.  R <= 2048, P <= 64. */
typedef struct ekaid_spk_step_a_fwd {
    int B, R, A, P;
    const float* gxa;                 /* [B][4R] input-side gate pre-activations (bias_ih included) */
    ekaid_slabs_t gates_src[3];       /* hidden-side gate products, [B][4R] each */
    int n_gates_src;
    const float* b_hh;                /* [4R] */
    const float* c_prev;              /* [B][R] */
    float *h_out, *c_out, *gate_save; /* [B][R], [B][R], [B][4R] */
    const float *w_fc, *b_fc;         /* [3][R], [3] */
    const float *bef, *diff, *aft;    /* [B][A] */
    float *mw, *att, *att2;           /* [B][3], [B][A], strided copy (ld2) or NULL */
    int ld2;
    ekaid_slabs_t pos_src;            /* pos1 product [B][R] */
    const float *b_p1, *w_wp, *b_wp, *w_p2, *b_p2;     /* [R], [P][R], [P], [R][P], [R] */
    float p_v; uint64_t seed_v; float p_d; uint64_t seed_d;
    float *vp, *dp, *pw, *pp;         /* [B][R], [B][P], [B][P], strided (ldpp) */
    int ldpp;
} ekaid_spk_step_a_fwd_t;
int ekaid_spk_step_a_fwd(const ekaid_spk_step_a_fwd_t* d, void* stream);

/* The same chain backwards (autograd of :102-121), one workgroup per batch row:
 *   1. d_att[k] = sum(dpa_src)[R + k] + d_gated * gate;  d_bef/d_diff/d_aft += mw_c * d_att;  dz_fc = softmax'(mw) <d_att, feat_c>
 *   2. att-LSTM cell: dh = sum(dh_carry) + dz_fc . w_fc; dc = dc_in -> dgates (leading dim lddg), dc_out
 *   3. POS chain from dpp = sum(dpa_src)[k < R] (also written to dpp_out, leading dim lddpp) -> dwp_out, dx_out (lddx);
 *      dhead (may be NULL) is added at dpos. */
typedef struct ekaid_spk_step_a_bwd {
    int B, R, A, P;
    ekaid_slabs_t dpa_src;            /* [B][R + A]: d [pos_feat | att (gate-MLP path)] */
    const float *d_gated, *gate;      /* [B][A] */
    const float *mw, *bef, *diff, *aft, *w_fc;
    float *dz_fc, *d_bef, *d_diff, *d_aft;          /* [B][3]; [B][A] accumulated in place */
    ekaid_slabs_t dh_carry;           /* [B][R] gradient arriving at h_att from the following step (absent at the last step) */
    const float *dc_in, *gate_save, *c_prev, *c_cur;
    float* dgates; int lddg; float* dc_out;
    const float *pw, *dhead, *w_p2, *w_wp, *vp;
    float p_v; uint64_t seed_v; float p_d; uint64_t seed_d;
    float* dpp_out; int lddpp; float* dwp_out; float* dx_out; int lddx;
} ekaid_spk_step_a_bwd_t;
int ekaid_spk_step_a_bwd(const ekaid_spk_step_a_bwd_t* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sequence-level entry points: the whole time loop of a recurrent block behind ONE call (csrc/sequence.hip).  Each step is
 * the chain of slab products and slab-consuming kernels declared above, issued from C on `stream`; no allocation, no host
 * synchronisation (graph-capturable).  `workspace` holds the K-slabs of one step and the small carries of the backward
 * loops; its size comes from the matching *_workspace_bytes call (backward = 0 / 1).
 *
 * Question GRU, 1 layer, gate order r,z,n, h0 = hall[0] (ref: language_model.py:106-115, nn.GRU):
 *   gi   [B][T][3H]  x W_ih^T + b_ih for all steps (one time-batched ekaid_gemm_f32 by the caller)
 *   hall [T+1][B][H] time-major states, row 0 given, rows 1..T written;  save [T][B][4H] gate activations for the backward
 *   bwd: dout [B][T][H] -> dgi [B][T][3H], dgh [T][B][3H]; the weight / bias gradients are time-batched products of the
 *   caller (dW_hh = dgh^T hall[0:T], dW_ih = dgi^T x, a_rowsum for the biases). */
size_t ekaid_gru_seq_workspace_bytes(int B, int T, int H, int backward);
int ekaid_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, int B, int T, int H, float* hall, float* save,
                      void* workspace, size_t workspace_bytes, void* stream);
int ekaid_gru_seq_bwd(const float* dout, const float* w_hh, const float* hall, const float* save, int B, int T, int H, float* dgi,
                      float* dgh, void* workspace, size_t workspace_bytes, void* stream);
/* The same two sequences as ONE persistent launch each (csrc/persist_gru.hip; H = 1024 only, otherwise EKAID_ERR_UNSUPPORTED and a
 * workspace query of 0): one seam + one 16 x 16 MFMA job per step instead of two launches.  Same arguments and results (to f32
 * rounding; deterministic).  ekaid_gru_seq_fwd / _bwd route here under ekaid_set_seq_mode(1) when the shape is supported;
 * ekaid_gru_seq_workspace_bytes covers either form. */
size_t ekaid_gru_seq_persistent_workspace_bytes(int B, int T, int H, int backward);
int ekaid_gru_seq_fwd_persistent(const float* gi, const float* w_hh, const float* b_hh, int B, int T, int H, float* hall, float* save,
                                 void* workspace, size_t workspace_bytes, void* stream);
int ekaid_gru_seq_bwd_persistent(const float* dout, const float* w_hh, const float* hall, const float* save, int B, int T, int H,
                                 float* dgi, float* dgh, void* workspace, size_t workspace_bytes, void* stream);

/* Teacher-forced answer decoder, the recurrent part of DynamicSpeaker._forward (ref: dynamic_speaker_change_pos.py:94-131,
 * 196-222): per step module-attention LSTM -> module weights + attended feature -> POS chain -> gate MLP -> language LSTM;
 * 8 launches per step in each direction.  Time-batched around it by the caller: embeddings, input-side gates GXA / GXL
 * (bias_ih included), logits / log-softmax heads, all weight and bias gradients (K = T*B products over the saved tensors).
 * Layout of the NW = 9R + G1 wide "h_lang projection" block (Wcat rows, DCAT columns):
 *   [0,4R) W_ih_att[:, E:] | [4R,5R) W_pos1 | [5R,5R+G1) W_g1[:, :R] | [5R+G1,NW) W_hh_lang.
 * All activations time-major, contiguous. */
typedef struct ekaid_speaker_seq {
    int T, B, R, A, P, G1, We;        /* steps, batch, rnn size, feature width, POS classes, gate-MLP width, word-embedding width */
    float p_lm, p_pos;                /* dropout probabilities (0 in eval mode) */
    const uint64_t* seeds;            /* HOST array [T][3]: seeds of the vpos, dpos and gate-MLP dropouts of each step (NULL if p = 0) */
    const float *Wcat, *whh_a, *bhh_a, *w_fc, *b_fc, *b_p1, *w_wp, *b_wp, *w_p2, *b_p2;
    const float *w_g1, *b_g1, *w_g2, *b_g2, *wih_l, *bhh_l;   /* w_g1 [G1][2R+A], w_g2 [A][G1], wih_l [4R][We+A] */
    const float *bef, *diff, *aft;    /* [B][A] */
    const float *GXA, *GXL;           /* [T][B][4R] */
    float *HA, *CA, *HL, *CL;         /* [T+1][B][R]: row 0 = initial state (given), rows 1..T written by fwd */
    float *SA, *SL;                   /* [T][B][4R] gate activations */
    float *G1D, *MW, *VP, *DP, *PW, *PA, *AT, *GT, *GA;   /* [T][B][G1], [..][3], [..][R], [..][P], [..][P], [..][R+A], [..][A] x3 */
} ekaid_speaker_seq_t;
typedef struct ekaid_speaker_seq_bwd {
    ekaid_speaker_seq_t f;            /* the forward call's descriptor (weights and saved activations) */
    const float* dHLo;                /* [T][B][R] gradient reaching h_lang_t through the output head */
    const float* dDPl;                /* [T][B][P] gradient reaching the POS logits through their head, or NULL */
    float *DCAT, *DZFC, *DWP, *DPP, *DZ2;   /* out: [T][B][NW], [..][3], [..][P], [..][R], [..][A] (operands of the weight gradients) */
    float *d_bef, *d_diff, *d_aft;    /* [B][A], accumulated in place */
} ekaid_speaker_seq_bwd_t;
size_t ekaid_speaker_seq_workspace_bytes(const ekaid_speaker_seq_t* d, int backward);
int ekaid_speaker_seq_fwd(const ekaid_speaker_seq_t* d, void* workspace, size_t workspace_bytes, void* stream);
int ekaid_speaker_seq_bwd(const ekaid_speaker_seq_bwd_t* d, void* workspace, size_t workspace_bytes, void* stream);

/* The same forward sequence as ONE persistent launch (csrc/persist.hip): one workgroup per CU stays resident for all T steps, every
 * product is cut into 16-row x 16-column jobs with the full contraction (exact f32 on v_mfma_f32_16x16x4_f32, no split-K), cell
 * states stay in LDS, and the five all-to-all dependencies of a step are device-scope counter seams (write-through stores, sc1
 * loads) instead of kernel boundaries.  Same descriptor, same saved tensors (either forward pairs with either backward); results
 * agree with ekaid_speaker_seq_fwd to f32 rounding (different summation order), deterministic from run to run.
 * Supported: R = 512, A = 1024, P = 16, G1 = 2R + A, T <= 96, batch up to what the per-workgroup LDS state allows (256 on a 256-CU
 * part); anything else -> EKAID_ERR_UNSUPPORTED (workspace_bytes query returns 0).  `workspace` (256-byte aligned) starts with the
 * seam counters, zeroed by a memset in front of every launch; ekaid_speaker_seq_persistent_status_ptr() is a DEVICE pointer to the
 * per-device status word (0 = fine; see ekaid_seq_status_read).
 * ekaid_set_seq_mode(1) makes ekaid_speaker_seq_fwd / ekaid_speaker_greedy_steps route supported shapes here (0: step-wise chain;
 * default from EKAID_SEQ_MODE = stepwise | persistent, see csrc/api.hip).
 * ref: dynamic_speaker_change_pos.py:94-131 inside :196-222 */
size_t ekaid_speaker_seq_persistent_workspace_bytes(const ekaid_speaker_seq_t* d);
int ekaid_speaker_seq_fwd_persistent(const ekaid_speaker_seq_t* d, void* workspace, size_t workspace_bytes, void* stream);
const unsigned* ekaid_speaker_seq_persistent_status_ptr(const void* workspace);
/* The persistent sequence kernels never hang: a seam that does not open within ~0.5 s (e.g. because another process' kernels keep some
 * of the CUs, so that not all workgroups of the launch are resident) sets a per-device, STICKY status word and every workgroup leaves;
 * the outputs of that launch are garbage.  ekaid_seq_status_read copies the word to the host (synchronises `stream`): 0 = fine;
 * callers check it where they synchronise anyway (ekaid_amd: at log intervals, at the end of a run, in the tests). */
int ekaid_seq_status_read(unsigned* host_value, void* stream);
int ekaid_seq_status_clear(void* stream);
int ekaid_set_seq_mode(int mode);
int ekaid_get_seq_mode(void);

/* The backward sequence as ONE persistent launch (csrc/persist_bwd.hip), same descriptor and saved tensors as ekaid_speaker_seq_bwd.
 * Differences in what it returns: (i) DPP is NOT written -- the product d [pos_feat | att] = d gate1x . W_g1[:, R:] is never formed
 * (what it feeds is linear in the POS / module weights of a row); the caller obtains DPP, which only the pos2 weight gradient needs,
 * as ONE time-batched product DCAT[:, gate1x block] . W_g1[:, R:2R];  (ii) d_bef / d_diff / d_aft receive only the gradient that
 * reaches the features through the gating product; the part through gate-MLP layer 1 comes back as
 * dFm [3][B][G1] = sum_t mw_m[t] * d gate1x[t], and the caller adds dFm[m] . W_g1[:, 2R:] (one product per feature).
 * `workspace` (256-byte aligned, *_workspace_bytes(&b->f); 0 = unsupported shape) holds the seam counters, transposed weight copies
 * (made by the call), the hand-off buffers and the linear maps F_m / Wp of the forward.  Status word as for the forward
 * (ekaid_speaker_seq_persistent_status_ptr; 0x200 + seam).  Supported shapes as ekaid_speaker_seq_fwd_persistent.
 * ref: autograd of dynamic_speaker_change_pos.py:94-131 inside :196-222 */
size_t ekaid_speaker_seq_bwd_persistent_workspace_bytes(const ekaid_speaker_seq_t* d);
int ekaid_speaker_seq_bwd_persistent(const ekaid_speaker_seq_bwd_t* b, float* dFm, void* workspace, size_t workspace_bytes, void* stream);

/* Greedy decoding (sample_max = 1, eval mode) of steps t0 .. t1-1 without a host round trip per step: token embedding, word-side
 * gates, the recurrent step, logits, log-softmax and ekaid_greedy_step per step, the chosen tokens feeding the next step on
 * the device.  The caller polls n_unfinished[t] (samples still running after step t) every few steps and stops once it is 0.
 * ref: DynamicSpeaker._sample, dynamic_speaker_change_pos.py:287-357 (greedy branch), test_mimic.py:119-122.
 *   f        weights as for ekaid_speaker_seq_fwd, p_lm = p_pos = 0, T = the maximum number of steps (seq_length);
 *            f.HA / CA / HL / CL [T+1][B][R] (row 0 = initial state), f.MW [T][B][3] (kept per step), f.GXA [B][4R] = the
 *            embed-side gates of the attention LSTM (constant over the steps without dropout), f.GXL [B][4R] scratch, every
 *            other activation buffer of f ONE step large */
typedef struct ekaid_speaker_greedy {
    ekaid_speaker_seq_t f;
    const float* emb;                 /* token table [vocabulary + 1][We] */
    const float *bih_l, *w_lo, *b_lo; /* language-LSTM input bias [4R], logit layer [V][R], [V] */
    int V;
    float *xt, *logits, *logp;        /* scratch [B][We], [B][V], [B][V] */
    int64_t* it;                      /* [B] current tokens: <bos> ids before step 0, updated every step */
    int* unfinished;                  /* [B], zero before step 0 */
    int64_t* seq; float* seq_logp;    /* [B][seq_len], zero-filled by the caller */
    int seq_len;
    int* n_unfinished;                /* [seq_len], zero-filled by the caller */
} ekaid_speaker_greedy_t;
size_t ekaid_speaker_greedy_workspace_bytes(const ekaid_speaker_greedy_t* g);
int ekaid_speaker_greedy_steps(const ekaid_speaker_greedy_t* g, int t0, int t1, void* workspace, size_t workspace_bytes, void* stream);
/* The same range of steps as ONE persistent launch (csrc/persist.hip: the forward sequence kernel with three more stations per step:
 * word-side gates from the fed-back tokens, logits, row-local log-softmax / arg-max / stop flags).  Production widths only (workspace
 * query 0 otherwise); ekaid_speaker_greedy_steps routes ranges of >= 4 steps here under ekaid_set_seq_mode(1), and
 * ekaid_speaker_greedy_workspace_bytes covers either form. */
size_t ekaid_speaker_greedy_persistent_workspace_bytes(const ekaid_speaker_greedy_t* g, int steps);
int ekaid_speaker_greedy_steps_persistent(const ekaid_speaker_greedy_t* g, int t0, int t1, void* workspace, size_t workspace_bytes, void* stream);

/* Scheduled sampling, one decode step for the whole batch: it[b] = teacher[b * ld_teacher] with probability 1 - ss_prob, else a
 * draw from exp(logp_prev[b][:]) (counter-based uniforms from `seed`; the reference draws with torch.multinomial: same
 * distribution, its own random stream).   ref: dynamic_speaker_change_pos.py:197-208, schedule train_mimic.py:194-202 */
int ekaid_ss_tokens(const float* logp_prev, const int64_t* teacher, int ld_teacher, int B, int V, float ss_prob, uint64_t seed,
                    int64_t* it, void* stream);

/* out[r][c] = dropout(act(x[r*ldx + c] + bias[c]))   (contiguous out; mask index r*cols + c)
 * drop_act_bwd: out[r*ldo + c] = dy[r][c] * keep(seed, r*cols + c)/(1-p) * act'(y[r*ldy + c])  (y = activation output,
 * possibly after the same dropout: a dropped element has zero gradient either way) */
int ekaid_bias_act_drop(const float* x, int ldx, const float* bias, int rows, int cols, int act, float p, uint64_t seed,
                        float* out, void* stream);
int ekaid_drop_act_bwd(const float* dy, const float* y, int ldy, int rows, int cols, int act, float p, uint64_t seed,
                       float* out, int ldo, void* stream);

/* row softmax / log-softmax.  ref: dynamic_speaker_change_pos.py:61-63,71,238-239, modules.py:65 */
int ekaid_softmax_fwd(const float* x, float* y, long long rows, int cols, int log_mode, void* stream);
int ekaid_softmax_bwd(const float* y, const float* dy, float* dx, long long rows, int cols, int log_mode, void* stream);

/* att_feat = w0*a + w1*b + w2*c per row.  ref: dynamic_speaker_change_pos.py:112-117 */
int ekaid_mix3_fwd(const float* w, const float* a, const float* b, const float* c, int B, int C, float* out, void* stream);
int ekaid_mix3_bwd(const float* w, const float* a, const float* b, const float* c, const float* dout, int B, int C,
                   float* dw, float* da, float* db, float* dc, int accumulate /* 1: da/db/dc += */, void* stream);

/* masked NLL of teacher-forced log-probs.   ref: utils/utils.py:208-216
 * sums[0] = -sum logp[b][t][target[b][t]] * mask[b][t], sums[1] = sum mask  (t < T), both written by the call (fixed-order sums);
 * loss_out[0] = sums[0] / sums[1].  denom_dev (optional device scalar): overrides sums[1] -- under data parallelism
 * the caller passes (mask sum of the GLOBAL batch) / world_size, so that averaging the ranks' gradients reproduces the
 * loss of the global batch when the ranks hold answers of different lengths.
 * bwd: dlogp[b][t][v] = -(v == target) * mask * gscale_dev[0] / sums[1]   (dlogp zero elsewhere, written in full) */
int ekaid_nll_fwd(const float* logp, const int64_t* target, int ld_target, const float* mask, int ld_mask, int B, int T,
                  int V, const float* denom_dev, float* sums, float* loss_out, void* stream);
int ekaid_nll_bwd(const int64_t* target, int ld_target, const float* mask, int ld_mask, const float* sums,
                  const float* gscale_dev, int B, int T, int V, float* dlogp, void* stream);

/* out[0] = speaker_loss[0] + coef * (sum(att_bef) + sum(att_aft)), both of n elements.  ref: train_mimic.py:246-247
 * fill_scaled: out[i] = scalar_dev[0] * coef (its backward). */
int ekaid_total_loss_fwd(const float* speaker_loss, const float* att_bef, const float* att_aft, long long n, float coef,
                         float* out, void* stream);
int ekaid_fill_scaled(const float* scalar_dev, float coef, float* out, long long n, void* stream);

/* LayerNorm over the last dim (eps inside sqrt).  ref: modules.py:33,76 (eps = 1e-6). save: [rows][2] = mean, rstd */
int ekaid_layernorm_fwd(const float* x, const float* gamma, const float* beta, long long rows, int cols, float eps,
                        float* y, float* save, void* stream);
int ekaid_layernorm_bwd(const float* x, const float* gamma, const float* save, const float* dy, long long rows, int cols,
                        float* dx, float* dgamma, float* dbeta, void* stream);

/* integer-valued label matrix [B][S][S] -> one-hot float32 [B][N][N][L], channel l-1 = (label == l), cropped to the
 * first N nodes.  label_type: 0 = float64 (as the reference collates it), 1 = int64 (as stored in the HDF5 file),
 * 2 = int8 (the compact form the data pipeline ships over PCIe).   ref: utils/mimic_utils.py:119-149 */
int ekaid_process_matrix(const void* adj, int label_type, int B, int S, int N, int L, float* out, void* stream);

/* One greedy decode step for the whole batch (sample_max = 1).   ref: dynamic_speaker_change_pos.py:316-352
 * logp [B][V]; forbids token 0 at t == 0; it/seq[:, t]/seq_logp[:, t] written; unfinished[B] updated in place;
 * n_unfinished[0] (caller zero-fills) += samples still running -> the host may poll it every few steps instead of
 * synchronising every step. */
int ekaid_greedy_step(const float* logp, int B, int V, int t, int* unfinished, int64_t* it, int64_t* seq, float* seq_logp,
                      int seq_len, int* n_unfinished, void* stream);

/* dst[c][r] = src[r][c] (R x C, leading dims in elements).  Used once per weight per step: the forward GEMM reads
 * W^T [in][out] with 512-byte row segments instead of W [out][in] with 128-byte segments (measured +15-20 %). */
int ekaid_transpose(const float* src, int R, int C, int ld_src, float* dst, int ld_dst, void* stream);

/* Debug aid (EKAID_CHECK_FINITE=1 in ekaid_amd/ops.py): count[0] += number of Inf / NaN values among x[0..n); the caller
 * zero-fills count and reads it back.  ref: the reference's `assert not x.isnan().any()` after every layer, fc.py:42-48,
 * graph_att_layer.py:71-177 passim, language_model.py:111-152 (one host synchronisation each; off by default here). */
int ekaid_nonfinite_count(const float* x, long long n, int* count, void* stream);

/* Adam, torch.optim.Adam semantics (no amsgrad, wd = 0), over one flat buffer.   ref: utils/utils.py:96-99
 * g is multiplied by grad_scale first (1/world_size after an all-reduce sum). */
int ekaid_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, int step, float grad_scale, void* stream);
/* The same update with the step counter and the hyper-parameters in DEVICE memory, for hipGraph capture (a captured
 * by-value launch would freeze the bias correction and the learning-rate schedule): step_dev[0] is incremented by the
 * call, hyper_dev = {lr, beta1, beta2, eps, grad_scale}, derived_dev = 4 floats of scratch.
 * Fault guard (ABI 8): the update is skipped ON THE DEVICE -- step counter, moments and weights untouched -- when the per-device
 * status word of the persistent sequence kernels is set (a seam timed out: that step's gradients are garbage) or when
 * fault_flag (nullable) holds a non-zero value: the slot behind the flat gradient buffer into which every rank writes its own
 * status with ekaid_seq_fault_flag before the last bucket's all-reduce, so that ALL ranks skip a step one rank could not finish.
 * The host still raises where it reads the status word (ekaid_seq_status_read); nothing is applied in between. */
int ekaid_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper_dev, int* step_dev,
                        float* derived_dev, const float* fault_flag, void* stream);
/* flag[0] = 1.0f if the status word of the persistent sequence kernels is set on this device, else 0.0f (one tiny launch,
 * graph-capturable).  No reference counterpart: the reference has neither persistent kernels nor data parallelism. */
int ekaid_seq_fault_flag(float* flag, void* stream);

/* Launch counters (ABI 9): how many times this process has issued each of the routed kernels since the library was loaded -- host-side
 * atomics, bumped by the entry point that launches (a launch recorded into a hipGraph counts once, at capture).  They let a caller --
 * the parity tests -- prove WHICH implementation served a call made under the default routing (persistent sequence kernels vs the
 * step-wise chains; which GEMM arithmetic).  Unknown index: -1.  No reference counterpart. */
#define EKAID_COUNT_SPK_FWD_PERSISTENT 0    /* ekaid_speaker_seq_fwd_persistent (also when ekaid_speaker_seq_fwd routed there) */
#define EKAID_COUNT_SPK_BWD_PERSISTENT 1    /* ekaid_speaker_seq_bwd_persistent */
#define EKAID_COUNT_GREEDY_PERSISTENT 2     /* ekaid_speaker_greedy_steps_persistent (also when ekaid_speaker_greedy_steps routed there) */
#define EKAID_COUNT_GRU_FWD_PERSISTENT 3    /* ekaid_gru_seq_fwd_persistent */
#define EKAID_COUNT_GRU_BWD_PERSISTENT 4    /* ekaid_gru_seq_bwd_persistent */
#define EKAID_COUNT_SPK_FWD_STEPWISE 5      /* ekaid_speaker_seq_fwd served by the launch-per-product chain */
#define EKAID_COUNT_SPK_BWD_STEPWISE 6      /* ekaid_speaker_seq_bwd */
#define EKAID_COUNT_GREEDY_STEPWISE 7       /* ekaid_speaker_greedy_steps served by the chain */
#define EKAID_COUNT_GEMM_F32 8              /* products launched on the f32 matrix pipe (gemm mode 0, or a shape routed there) */
#define EKAID_COUNT_GEMM_SPLIT3 9           /* products launched as the exact 3-way bf16 split (6 matrix products per product) */
#define EKAID_COUNT_GEMM_SPLIT2 10          /* products launched as a 2-term split (3 matrix products per product) */
#define EKAID_COUNT_GAT_BWD_CORE 11        /* ekaid_gat_bwd_core launches (the graph-attention backward in its one-launch form) */
#define EKAID_COUNT_N 12
long long ekaid_launch_count(int which);

#ifdef __cplusplus
}
#endif
#endif
