// f32 GEMM on the CDNA4 matrix cores, three arithmetics behind one entry point (ekaid_gemm_f32):
//   gemm_split_kernel<.., NP = 2>: two-term f16 split with per-row / per-column power-of-two scales, three v_mfma_f32_32x32x16_f16
//                                products per f32 product -- the default for products that carry operand bounds (a_amax / b_amax);
//   gemm_split_kernel<.., NP = 3>: exact 3-way bf16 split of every operand element, six v_mfma_f32_32x32x16_bf16 products per f32
//                                product -- every other product (EKAID_GEMM_MODE=bf16x3: all of them);
//   gemm_mfma_kernel           : v_mfma_f32_32x32x2_f32, an exact k-ordered f32 fma chain (EKAID_GEMM_MODE=f32).
// All share the tiling, the operand loaders, the split-K planner and the epilogue; f32 accumulation throughout.
//
// C[M,N] = epilogue( alpha * op(A)[M,K] . op(B)[K,N] )     row-major everywhere
//   transA = 0 : A stored [M][K] (k-contiguous)      transA = 1 : A stored [K][M] (m-contiguous)
//   transB = 1 : B stored [N][K] (k-contiguous)      transB = 0 : B stored [K][N] (n-contiguous)
// The three forms the path needs (reference fc.py / nn.Linear call sites):
//   forward  y  = x . W^T      (transA=0, transB=1)
//   dgrad    dx = dy . W       (transA=0, transB=0)
//   wgrad    dW = dy^T . x     (transA=1, transB=0)
//
// Tiling: workgroup = 256 threads = 4 wave64 in a 2x2 grid; block tile (64*TM)x(64*TN), wave tile
// (32*TM)x(32*TN) as TMxTN MFMA tiles of 32x32 (16 accumulator VGPRs each).  K-step 32.
// Operand staging: global -> registers (16-B loads along the contiguous axis, issued before the
// MFMA phase of the previous K-step) -> LDS.  k-contiguous operands sit in LDS as [row][32+4]
// and are fetched with one ds_read_b128 per lane per 8-k group (lane half h supplies k = 4h..4h+3,
// the contraction order inside the group is permuted identically for A and B, which is exact
// reordering of an f32 sum); m/n-contiguous operands sit as [k][cols+4] and are fetched with
// ds_read_b32 (consecutive lanes -> consecutive banks).
// Small grids are split along K into `splitk` slabs reduced by a second kernel that also applies
// the epilogue (deterministic; no float atomics).
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <type_traits>
#include <cstdlib>

namespace ekaid {

#define BK_BIG 32      // K-step of the 128x128 kernel
#ifndef BK_HALF
#define BK_HALF 32     // K-step of the 128x64 / 64x128 kernels
#endif
#ifndef BK_SMALL
#define BK_SMALL 32    // K-step of the 64x64 / 32x32 kernels: their loop is load-latency bound (16 MFMAs per wave per
                       // 32-deep step), so one iteration carries 4x the bytes in flight (measured: 50 -> 90 TF at M = 256)
#endif
#define LDS_PAD 4

struct Epi {
    float alpha;
    const float* alpha_dev;
    int beta;
    const float* bias;
    const float* R;
    int ldr;
    int rdiv;
    const float* rowmask;
    int act;
    float drop_p, drop_inv;     // fused dropout of the activated output: mask index = m * N + n
    unsigned long long drop_seed;
    const unsigned long long* drop_epoch;   // device word folded into drop_seed (ekaid_set_dropout_epoch_ptr) or NULL
    int n_cols;
    const float* post;          // added AFTER the dropout (a gradient that joins the product: ekaid_gemm_t.post), leading dim ldpost
    int ldpost;
};

// cold = the old C element (e.beta), postv = the ekaid_gemm_t.post element (e.post); the callers fetch both for a whole group of
// elements BEFORE the first store of the group: a load issued between two stores cannot be moved above the earlier one (C and post may
// alias as far as the compiler knows), and an epilogue of 64 load -> add -> store round trips per lane costs more than the n-ary sum
// kernel the post operand replaces (measured: +37 us per launch at 18432 x 1024)
// The same holds for every other operand: `pre` = bias[n] + rowmask[m] * R[m / rdiv][n] and `seed` = the dropout seed with the epoch word
// folded in (a device word under hipGraph replay) are formed by the callers ahead of the stores as well (epi_pre / epi_seed).
__device__ __forceinline__ float epi_pre(const Epi& e, int m, int n) {
    float p = e.bias ? e.bias[n] : 0.f;
    if (e.R) {
        float r = e.R[(long long)(m / e.rdiv) * e.ldr + n];
        if (e.rowmask) r *= e.rowmask[m];
        p += r;
    }
    return p;
}
__device__ __forceinline__ uint64_t epi_seed(const Epi& e) { return e.drop_p > 0.f ? drop_seed_value(DropSeed{e.drop_seed, e.drop_epoch}) : 0ull; }
__device__ __forceinline__ float epilogue(float acc, float cold, float pre, float postv, uint64_t seed, int m, int n, const Epi& e, float alpha) {
    float v = acc * alpha;
    if (e.beta) v += cold;
    v += pre;
    v = act_apply(v, e.act);
    if (e.drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)m * e.n_cols + n, e.drop_p, e.drop_inv);
    return v + postv;
}

constexpr int ASUM_SHARE = 4;      // tile columns that share the A-row-sum work of a K-split product (partials per slab)
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    int nb2;
    int k_per_split;      // multiple of BK; == K rounded up when splitk == 1
    int splitk;
    float* ws;            // slabs [splitk][M][N] when splitk > 1
    int tiles_m, tiles_n;
    int slab_mode;        // 1: raw partial products go to ws even when splitk == 1 (ekaid_gemm_slabs: the consumer reduces)
    float* asum;          // optional (m-contiguous A only): asum[m] += sum_k op(A)[m][k] -- the bias gradient riding on a wgrad product
    float* asum_ws;       // its K-split partials [splitk][M] (behind the slabs in the workspace), finished by the reduce kernel
    const float* a_amax;  // device vectors: a_amax[m] >= max_k |op(A)[m][k]|, b_amax[n] >= max_k |op(B)[k][n]| (ekaid_gemm_t.a_amax / b_amax); the
    const float* b_amax;  // two-term f16 split scales every row of op(A) / column of op(B) by an exact power of two derived from them (NP = 2 only)
    Epi e;
};

// Operand staging is branch-free: every lane always issues its loads from a CLAMPED (in-bounds) address and
// out-of-range elements are zeroed with selects afterwards, so the loads of a K-step go out back-to-back
// and their latency hides under the MFMAs of the previous step (a per-element "load or zero" branch makes
// hipcc wait vmcnt(0) per element).  `vec` (16-byte loads legal: leading dim, base pointer and the extent
// along the contiguous axis all multiples of 4 elements) is kernel-uniform.
template <int NV>
struct TileLoader {
    const float* base[NV];   // per staged vector (up to 8: 64x64 tile with a 128-deep K-step): pointer to its (clamped) row / k-row start + column offset
    int lim[NV];             // KC: row valid (0/1);  MC: number of valid elements along the contiguous axis (<=4, may be <=0)
};

template <bool KC, int BT, int NV, int KB>
__device__ __forceinline__ void loader_init(TileLoader<NV>& L, const float* P, int ld, int t0, int T, int K, int tid, bool vec) {
    // KC: tile rows t (extent T) are matrix rows, contiguous axis = k.   MC: contiguous axis = t, rows = k.
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (KC) {
            const int row = idx / (KB / 4), c4 = idx % (KB / 4);
            const int gt = t0 + row;
            L.base[i] = P + (long long)min(gt, T - 1) * ld + 4 * c4;
            L.lim[i] = gt < T ? 1 : 0;
        } else {
            const int c4 = idx % (BT / 4);
            const int gt = t0 + 4 * c4;
            const int gtc = vec ? min(gt, T - 4) : min(gt, T - 1);
            L.base[i] = P + max(gtc, 0);
            L.lim[i] = T - gt;
        }
    }
}

// issue the loads of one K-step (no consumer of the loaded values here: the zero-masking happens in
// loader_mask(), called after the MFMA phase right before the LDS store, so nothing waits on vmcnt early)
template <bool KC, int BT, int NV, int KB, bool FASTVEC>
__device__ __forceinline__ void loader_fetch(const TileLoader<NV>& L, f32x4* r, int ld, int k0, int K, int T, int t0, int tid,
                                             bool vec_rt) {
    const bool vec = FASTVEC ? true : vec_rt;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (KC) {
            const int c4 = idx % (KB / 4);
            const int gk = k0 + 4 * c4;
            if (vec) {
                r[i] = *reinterpret_cast<const f32x4*>(L.base[i] + (min(gk, K - 4) - 4 * c4));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = L.base[i][min(gk + e, K - 1) - 4 * c4];
            }
        } else {
            const int krow = idx / (BT / 4);
            const float* p = L.base[i] + (long long)min(k0 + krow, K - 1) * ld;
            if (vec) {
                r[i] = *reinterpret_cast<const f32x4*>(p);
            } else {
                const int c4 = idx % (BT / 4);
                const int gt = t0 + 4 * c4;
                const int gtc = max(min(gt, T - 1), 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[i][e] = p[min(gt + e, T - 1) - gtc];
            }
        }
    }
}

template <bool KC, int BT, int NV, int KB>
__device__ __forceinline__ void loader_mask(const TileLoader<NV>& L, f32x4* r, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (KC) {
            const int gk = k0 + 4 * (idx % (KB / 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) r[i][e] = (L.lim[i] && gk + e < kend) ? r[i][e] : 0.f;
        } else {
            const bool kok = (k0 + idx / (BT / 4)) < kend;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[i][e] = (kok && e < L.lim[i]) ? r[i][e] : 0.f;
        }
    }
}

// epilogue shared by the MFMA kernels.  C/D map of the 32x32 MFMA: col = lane&31, row = (t&3) + 8*(t>>2) + 4*(lane>>5)
// sh (two-term f16 split only, else nullptr): LDS table of the power-of-two shifts the tile's BM rows and BN columns were scaled by; every
// accumulator leaves the scaled domain as it is consumed (exact).  Read tile by tile: hoisted ahead of the stores, the 32 + 2 shifts of a
// lane cost more registers than the 128x128 pair kernels have left (49 spilled VGPRs, 1500 scratch instructions: measured)
template <int TM, int TN, bool RESCALE = false>
__device__ __forceinline__ void write_tiles(const GemmArgs& g, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int r, int h, int z,
                                            const int* sh = nullptr) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const bool slab = g.splitk > 1 || g.slab_mode;
    float* C;
    int ldc;
    if (slab) {
        C = g.ws + (long long)z * g.M * g.N;
        ldc = g.N;
    } else {
        const long long z1 = z / g.nb2, z2 = z % g.nb2;
        C = g.C + z1 * g.sC1 + z2 * g.sC2;
        ldc = g.ldc;
    }
    const float alpha = g.e.alpha_dev ? g.e.alpha * g.e.alpha_dev[0] : g.e.alpha;
    const uint64_t seed = slab ? 0ull : epi_seed(g.e);
    const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    auto store_tiles = [&](auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const int n = n0 + wn + 32 * jj + r;
                if (CHECK && n >= g.N) continue;
                const int mb = m0 + wm + 32 * i + 4 * h;
                float* cp0 = C + (long long)mb * ldc + n;
                float cold[16], postv[16], pre[16];              // the operands of this lane's 16 elements, all requested before the first store
                f32x16 tile = acc[i][jj];
                if constexpr (RESCALE) {
                    const int sb = sh[BM + wn + 32 * jj + r];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                // rows 4h + 8q .. + 3 of the tile: four consecutive table entries
                        const int4 sa = *reinterpret_cast<const int4*>(&sh[wm + 32 * i + 4 * h + 8 * q]);
                        tile[4 * q + 0] = ldexpf(tile[4 * q + 0], -(sa.x + sb));
                        tile[4 * q + 1] = ldexpf(tile[4 * q + 1], -(sa.y + sb));
                        tile[4 * q + 2] = ldexpf(tile[4 * q + 2], -(sa.z + sb));
                        tile[4 * q + 3] = ldexpf(tile[4 * q + 3], -(sa.w + sb));
                    }
                }
                const bool has_cold = !slab && g.e.beta, has_post = !slab && g.e.post, has_res = !slab && g.e.R;
                const float* pp0 = has_post ? g.e.post + (long long)mb * g.e.ldpost + n : nullptr;
                const float bias = (!slab && g.e.bias) ? g.e.bias[n] : 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int dm = (t & 3) + 8 * (t >> 2);
                    const bool ok = !CHECK || mb + dm < g.M;
                    cold[t] = (has_cold && ok) ? cp0[(long long)dm * ldc] : 0.f;
                    postv[t] = (has_post && ok) ? pp0[(long long)dm * g.e.ldpost] : 0.f;
                    pre[t] = (has_res && ok) ? epi_pre(g.e, mb + dm, n) : bias;
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int dm = (t & 3) + 8 * (t >> 2);
                    if (CHECK && mb + dm >= g.M) continue;
                    float* cp = cp0 + (long long)dm * ldc;
                    *cp = slab ? tile[t] : epilogue(tile[t], cold[t], pre[t], postv[t], seed, mb + dm, n, g.e, alpha);
                }
            }
    };
    if (full) store_tiles(std::false_type{});
    else store_tiles(std::true_type{});
}

#ifndef EKAID_SPLIT_OCC_BIG
#define EKAID_SPLIT_OCC_BIG 3
#endif
#ifndef EKAID_SPLIT_ABLATE
#define EKAID_SPLIT_ABLATE 0
#endif
#ifndef EKAID_SPLIT_PIPE_BIG
#define EKAID_SPLIT_PIPE_BIG true
#endif
#ifndef EKAID_SPLIT_OCC_SMALL
#define EKAID_SPLIT_OCC_SMALL 4
#endif
#ifndef EKAID_SPLIT_KB_SMALL
#define EKAID_SPLIT_KB_SMALL 32
#endif
#ifndef EKAID_SPLIT_PF_SMALL
#define EKAID_SPLIT_PF_SMALL 1
#endif
#ifndef EKAID_SPLIT_PIPE_SMALL
#define EKAID_SPLIT_PIPE_SMALL false      // software-pipelined K loop for the 64x64 tile (build-time experiment)
#endif
#ifndef EKAID_LB_HALF
#define EKAID_LB_HALF 4
#endif
// workgroups per CU the register allocator must admit: 3 for the 128x128 tile (<= 168 VGPR+AGPR), more for smaller tiles
template <bool A_KC, bool B_KC, int TM, int TN, int KB>
__global__ __launch_bounds__(256, (TM * TN >= 4 ? 3 : (TM * TN == 2 ? EKAID_LB_HALF : 6))) void gemm_mfma_kernel(GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_ELEMS = A_KC ? BM * (KB + LDS_PAD) : KB * (BM + LDS_PAD);
    constexpr int B_ELEMS = B_KC ? BN * (KB + LDS_PAD) : KB * (BN + LDS_PAD);
    constexpr int A_V4 = BM * KB / 4 / 256;     // float4 per thread per tile
    constexpr int B_V4 = BN * KB / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // A tile | B tile
    constexpr int STAGE = A_ELEMS + B_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN);

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give every
    // XCD a contiguous run of tiles -> neighbouring tiles (same A row panel) share one L2.
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int m0 = (bid / g.tiles_n) * BM, n0 = (bid % g.tiles_n) * BN;

    const int z = blockIdx.z;
    const float* A = g.A; const float* B = g.B;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = z * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    } else {
        const long long z1 = z / g.nb2, z2 = z % g.nb2;
        A += z1 * g.sA1 + z2 * g.sA2;
        B += z1 * g.sB1 + z2 * g.sB2;
    }
    // contiguous-axis extent: k for k-contiguous operands, m (A) / n (B) otherwise
    const bool vecA = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (((A_KC ? g.K : g.M) & 3) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (((B_KC ? g.K : g.N) & 3) == 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

    f32x4 ra[A_V4], rb[B_V4];
    TileLoader<A_V4> LA;
    TileLoader<B_V4> LB;
    loader_init<A_KC, BM, A_V4, KB>(LA, A, g.lda, m0, g.M, g.K, tid, vecA);
    loader_init<B_KC, BN, B_V4, KB>(LB, B, g.ldb, n0, g.N, g.K, tid, vecB);

    int kload = kbeg;       // K offset of the tile currently held in ra/rb
    // zero-masking is only needed for ragged K (rows/columns beyond M/N are computed on clamped data and never stored)
    const bool ragged = ((kend - kbeg) % KB) != 0 || !vecA || !vecB ||
                        (!A_KC && (m0 + BM > g.M)) || (!B_KC && (n0 + BN > g.N));
    auto lstore = [&](int buf, bool need_mask) {
        float* As = lds + buf * STAGE;
        float* Bs = As + A_ELEMS;
        if (need_mask) {
            loader_mask<A_KC, BM, A_V4, KB>(LA, ra, kload, kend, tid);
            loader_mask<B_KC, BN, B_V4, KB>(LB, rb, kload, kend, tid);
        }
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int idx = tid + i * 256;
            if (A_KC) {
                const int row = idx / (KB / 4), c4 = idx % (KB / 4);
                *reinterpret_cast<f32x4*>(&As[row * (KB + LDS_PAD) + 4 * c4]) = ra[i];
            } else {
                const int krow = idx / (BM / 4), c4 = idx % (BM / 4);
                *reinterpret_cast<f32x4*>(&As[krow * (BM + LDS_PAD) + 4 * c4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int idx = tid + i * 256;
            if (B_KC) {
                const int row = idx / (KB / 4), c4 = idx % (KB / 4);
                *reinterpret_cast<f32x4*>(&Bs[row * (KB + LDS_PAD) + 4 * c4]) = rb[i];
            } else {
                const int krow = idx / (BN / 4), c4 = idx % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[krow * (BN + LDS_PAD) + 4 * c4]) = rb[i];
            }
        }
    };
    // MFMA operand fragments of one 8-k group: k-contiguous tiles give them as one ds_read_b128 per lane
    // (lane half h holds k = 4h..4h+3), the others as four ds_read_b32 (consecutive lanes -> consecutive banks)
    auto lfrag = [&](const float* As, const float* Bs, int kb, f32x4* a, f32x4* b) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (A_KC) {
                a[i] = *reinterpret_cast<const f32x4*>(&As[(wm + 32 * i + r) * (KB + LDS_PAD) + kb + 4 * h]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[i][j] = As[(kb + 4 * h + j) * (BM + LDS_PAD) + wm + 32 * i + r];
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (B_KC) {
                b[i] = *reinterpret_cast<const f32x4*>(&Bs[(wn + 32 * i + r) * (KB + LDS_PAD) + kb + 4 * h]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[i][j] = Bs[(kb + 4 * h + j) * (BN + LDS_PAD) + wn + 32 * i + r];
            }
        }
    };

    // the K loop exists twice: FAST = every staged vector is a legal, fully in-range 16-byte load (no runtime
    // alignment test, no masking: straight-line loads), otherwise the general clamped + masked form
    auto mainloop = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        auto gload = [&](int k0) {
            kload = k0;
            loader_fetch<A_KC, BM, A_V4, KB, FAST>(LA, ra, g.lda, k0, g.K, g.M, m0, tid, vecA);
            loader_fetch<B_KC, BN, B_V4, KB, FAST>(LB, rb, g.ldb, k0, g.K, g.N, n0, tid, vecB);
        };
        gload(kbeg);
        lstore(0, !FAST);
        __syncthreads();
        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += KB) {
#if defined(EKAID_GEMM_ABLATE)
            const bool more = false;                  // timing-only ablation: no staging after the first tile
#else
            const bool more = (k0 + KB) < kend;
#endif
            if (more) gload(k0 + KB);                 // next tile's HBM/L2 latency hides under this step's MFMAs
            const float* As = lds + cur * STAGE;
            const float* Bs = As + A_ELEMS;
            f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int gidx = 0; gidx < KB / 8; ++gidx) {
                lfrag(As, Bs, 8 * gidx, fa[gidx & 1], fb[gidx & 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gidx & 1][i][j], fb[gidx & 1][jj][j], acc[i][jj], 0, 0, 0);
            }
            __syncthreads();
            if (more) { lstore(0, !FAST); __syncthreads(); }
        }
    };
    if (kbeg < kend) {
        if (ragged) mainloop(std::false_type{});
        else mainloop(std::true_type{});
    }

    write_tiles<TM, TN>(g, acc, m0, n0, wm, wn, r, h, z);
}

// ------------------------------------------------------------------------------------------------------------------
// f32 GEMM on the bf16 matrix pipe ("3-way exact split", 6 products).
//
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 MFMA rate.  An f32 value is EXACTLY the sum of three bf16
// values  a = a_h + a_m + a_l  (8 + 8 + 8 significant bits, obtained by truncation, every residual is exact in f32), so
//     a * b = a_h b_h + (a_h b_m + a_m b_h) + (a_m b_m + a_h b_l + a_l b_h) + [a_m b_l + a_l b_m + a_l b_l]
// where each kept product of two bf16 values is exact in the f32 accumulator of v_mfma_f32_32x32x16_bf16 and the
// dropped bracket is below 2^-23 |a b| -- the size of ONE f32 rounding of the product.  Six bf16 MFMAs (6 x 32 cycles
// per 32x32x16 block) replace eight f32 MFMAs (8 x 64 cycles): 2.67x the matrix throughput at f32-level accuracy
// (measured against float64 in tests/test_hip_ops.py::test_gemm_split_accuracy: same error as the f32 pipe).
//
// Staging: global f32 -> registers (same loaders as above) -> split at LDS-store time into three bf16 planes.
//   k-contiguous operand : plane image [row][KB] bf16; fragment = one ds_read_b128 per plane
//                          (lane (r, h) takes k = 8h .. 8h+7 of row r: the operand map of the 32x32x16 MFMA)
//   m/n-contiguous operand: plane image [k][rows] bf16 exactly as it lies in memory (8-byte stores, no transpose);
//                          fragment = two ds_read_b64_tr_b16 per plane (the LDS hardware transposes 4 k x 16 rows
//                          per 16-lane group)
// Both images are unpadded and XOR-swizzled (kc_store_off / mc_store_off below): 48 KiB for the 128x128 tile, so
// three workgroups share a CU.
#ifndef EKAID_SPLIT_ABLATE
#define EKAID_SPLIT_ABLATE 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// four f32 -> three planes of four bf16 (truncation split; h + m + l == v exactly for finite v).
// (A round-to-nearest split on v_cvt_pk_bf16_f32 + v_pk_add_f32 is 18 instead of 22 vector instructions per four words and
// equally exact, but the 64-bit register pairs of the packed subtracts push the 128x128 kernel over its 168-VGPR budget
// (62-179 spilled registers in the k-contiguous variants: 2.5x slower) and, applied only where it fits, measured no
// faster: 182 vs 187 TF on the wgrad form.)
// NP = 3: the exact split (h + m + l == v), six products per block.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// NP = 2: the two-term f16 split.  With s an exact power of two that brings the largest element of the element's row of op(A) /
// column of op(B) into [2^14, 2^15):
//   h = f16(v s) (round to nearest: 11 significant bits),  l = f16(v s - h) (the exact f32 residual, |l| <= 2^-11 |v s|, rounded to 11 more)
// so v s = h + l + e with |e| <= 2^-22 |v s| (a quarter of that on average) as long as l is a normal f16 -- elements within 2^-17 of the
// row's largest; smaller ones keep an ABSOLUTE accuracy of 2^-25 (f16 subnormals, which the matrix pipe does not flush), i.e.
// 2^-40 of the row's largest element.  Three products per block (h h + h l + l h on v_mfma_f32_32x32x16_f16): the dropped l l is
// below 2^-22 |a b|.  M receives l; L is unused.  s = the scales of the four elements (one row: all equal).
template <int NP = 3>
__device__ __forceinline__ void split4(const f32x4& v, const f32x4& s, u32x2& H, u32x2& M, u32x2& L) {
    if constexpr (NP == 2) {
        const f32x2 a0 = {v[0] * s[0], v[1] * s[1]}, a1 = {v[2] * s[2], v[3] * s[3]};
        const f16x2 h0 = __builtin_convertvector(a0, f16x2), h1 = __builtin_convertvector(a1, f16x2);
        const f32x2 r0 = {a0[0] - (float)h0[0], a0[1] - (float)h0[1]}, r1 = {a1[0] - (float)h1[0], a1[1] - (float)h1[1]};
        const f16x2 l0 = __builtin_convertvector(r0, f16x2), l1 = __builtin_convertvector(r1, f16x2);
        H[0] = __builtin_bit_cast(unsigned, h0); H[1] = __builtin_bit_cast(unsigned, h1);
        M[0] = __builtin_bit_cast(unsigned, l0); M[1] = __builtin_bit_cast(unsigned, l1);
        return;
    }
#if EKAID_SPLIT_ABLATE == 2              // timing-only: no split arithmetic (three copies of the raw words)
    H[0] = __float_as_uint(v[0]); H[1] = __float_as_uint(v[1]); M[0] = __float_as_uint(v[2]); M[1] = __float_as_uint(v[3]); L = H;
    return;
#endif
    unsigned u[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[e] & 0xFFFF0000u);
        m[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(m[e] & 0xFFFF0000u);
        l[e] = __float_as_uint(r2);
    }
    // pack the upper halves of two words: result = (x0 >> 16) | (x1 & 0xFFFF0000)
    H[0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u); H[1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    M[0] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302u); M[1] = __builtin_amdgcn_perm(m[3], m[2], 0x07060302u);
    L[0] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302u); L[1] = __builtin_amdgcn_perm(l[3], l[2], 0x07060302u);
}

// LDS plane images, unpadded with XOR swizzles so that every access below is bank-conflict free (K-step KB = 32/64/128):
//   k-contiguous: [row][KB/8 slots of 16 B]; slot s of row R sits at slot s ^ key(R), key(R) = (R / (128/KB)) mod (KB/8).
//                 ds_read_b128 serves 16 lanes (16 rows, distinct mod 16) per cycle; 2*KB-byte rows put 128/KB rows on
//                 one 256-B bank row, the key spreads the rows that share a bank row position over the slots.
//   m-contiguous: [k][BT/32 chunks of 64 B]; chunk ch of k-row k sits at ch ^ (k & 3) (128-wide tile, 256-B rows) or
//                 ch ^ ((k >> 1) & 1) (64-wide tile, 128-B rows): the 4 k-rows a transposed read gathers per half-wave
//                 land on 4 different 64-B bank groups.
template <int KB>
__device__ __forceinline__ int kc_key(int row) { return (row / (128 / KB)) & (KB / 8 - 1); }
template <int KB>
__device__ __forceinline__ int kc_store_off(int idx) {        // idx: row = idx / (KB/4), float4 c4 = idx % (KB/4) (k = 4 c4)
    const int row = idx / (KB / 4), c4 = idx % (KB / 4);
    return row * (2 * KB) + 16 * ((c4 >> 1) ^ kc_key<KB>(row)) + 8 * (c4 & 1);
}
template <int KB>
__device__ __forceinline__ int kc_read_off(int row, int slot) { return row * (2 * KB) + 16 * (slot ^ kc_key<KB>(row)); }
template <int BT>
__device__ __forceinline__ int mc_store_off(int idx) {        // idx: k-row = idx / (BT/4), float4 c4 (rows 4 c4 ..)
    const int k = idx / (BT / 4), c4 = idx % (BT / 4);
    const int key = BT == 128 ? (k & 3) : ((k >> 1) & 1);
    return k * (2 * BT) + 64 * ((c4 >> 3) ^ key) + 8 * (c4 & 7);
}
template <int BT>
__device__ __forceinline__ int mc_read_off(int k, int row0, int tg, int tp) {   // row0: first of the tile's 32 rows
    const int key = BT == 128 ? (k & 3) : ((k >> 1) & 1);
    return k * (2 * BT) + 64 * ((row0 >> 5) ^ key) + 32 * tg + 8 * tp;
}

// power of two that brings a tensor whose largest magnitude is `amax` into [2^14, 2^15) (f16 tops out at 65504; the headroom also
// covers the rounding of the largest element).  Zero, NaN and Inf bounds leave the operand as it is.
__device__ __forceinline__ int f16_shift(const float amax) {
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 0;
    int e;
    (void)frexpf(amax, &e);                    // amax = m 2^e, 0.5 <= m < 1
    const int s = 15 - e;
    return s < -126 ? -126 : (s > 126 ? 126 : s);
}

// PF = K-steps of global loads kept in flight (1: the next step's loads hide under this step's MFMAs -- enough for the
// large tiles; 2: for the 64x64 tile, whose 12 MFMAs per 32-deep step are far shorter than a load round trip)
template <bool A_KC, bool B_KC, int TM, int TN, int KB, int PF, bool PIPE, int NP = 3, bool ASUM = false>
__device__ __forceinline__ void gemm_split_body(const GemmArgs& g, const int block_x, const int z) {
    static_assert(NP == 2 || NP == 3, "planes per operand");
    static_assert(!ASUM || !A_KC, "the A row sums ride on the m-contiguous staging (weight-gradient form)");
    static_assert(!PIPE || (KB == 32 && PF == 1), "the pipelined loop shares the 32-deep LDS footprint");
    constexpr int BM = 64 * TM, BN = 64 * TN;
    static_assert(KB == 32 || KB == 64 || KB == 128, "K-step");
    static_assert(PF == 1 || PF == 2, "prefetch depth");
    constexpr int A_RS = A_KC ? 2 * KB : 2 * BM;          // row stride of one plane image, bytes (no padding: XOR swizzles)
    constexpr int B_RS = B_KC ? 2 * KB : 2 * BN;
    constexpr int A_PLANE = (A_KC ? BM : KB) * A_RS;
    constexpr int B_PLANE = (B_KC ? BN : KB) * B_RS;
    constexpr int A_BYTES = NP * A_PLANE;
    constexpr int A_V4 = BM * KB / 4 / 256;
    constexpr int B_V4 = BN * KB / 4 / 256;
    constexpr int NCH = KB / 16;                          // 16-k MFMA chunks per K-step
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const ldsA = reinterpret_cast<char*>(lds);
    char* const ldsB = ldsA + A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * (32 * TM), wn = (wave & 1) * (32 * TN);

    const int nwg = g.tiles_m * g.tiles_n;
    int bid = block_x;
    {
        const int q = nwg >> 3, rem = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int m0 = (bid / g.tiles_n) * BM, n0 = (bid % g.tiles_n) * BN;
    // ASUM: the workgroups of the first tile column also add up the A elements they stage (every staged float4 of a thread
    // covers the same four tile rows m, 256 % (BM / 4) == 0), workgroup-uniform branch
    // With a K split the K-steps are dealt round-robin to the workgroups of a tile row (each adds up 1 / tiles_n of the steps:
    // the extra vector work is spread evenly instead of making the first tile column the straggler) and every workgroup
    // leaves a partial [z][tile column][M] for the reduce kernel; without a split the first tile column does all of it.
    const int tcol = bid % g.tiles_n;
    const int asum_period = g.splitk > 1 ? min(g.tiles_n, ASUM_SHARE) : 1;      // workgroups of a tile row that take turns
    const bool do_asum = ASUM && g.asum != nullptr && tcol < asum_period;
    int asum_turn = tcol;                                 // K-steps until this workgroup's next turn
    auto asum_mine = [&]() { const bool mine = asum_turn == 0; asum_turn = mine ? asum_period - 1 : asum_turn - 1; return mine; };
    f32x4 asum4 = {0.f, 0.f, 0.f, 0.f};

    const float* A = g.A; const float* B = g.B;
    int kbeg = 0, kend = g.K;
    if (g.splitk > 1) {
        kbeg = z * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    } else {
        const long long z1 = z / g.nb2, z2 = z % g.nb2;
        A += z1 * g.sA1 + z2 * g.sA2;
        B += z1 * g.sB1 + z2 * g.sB2;
    }
    const bool vecA = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (((A_KC ? g.K : g.M) & 3) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (((B_KC ? g.K : g.N) & 3) == 0);
    // two-term f16 split: the exact power-of-two scale of row m of op(A) / column n of op(B), from the callers' bounds (rows / columns
    // past the matrix: 1).  A thread stages the same rows in every K-step, so its scales are loaded once, ahead of the loop.
    auto scale_of = [&](const float* amax, int idx, int extent) -> float {
        return (NP == 2 && idx < extent) ? ldexpf(1.f, f16_shift(amax[idx])) : 1.f;
    };
    constexpr int NSM = (NP == 2) ? 4 : 1;
    f32x4 scA_mc = {1.f, 1.f, 1.f, 1.f}, scB_mc = {1.f, 1.f, 1.f, 1.f};      // m/n-contiguous staging: a float4 = 4 rows, the same 4 for every float4 of a thread
    float scA_kc[A_KC ? NSM : 1], scB_kc[B_KC ? NSM : 1];                   // k-contiguous staging of the plain loop: one row per float4 (A_V4, B_V4 <= 4)
    static_assert(NP != 2 || (A_V4 <= 4 && B_V4 <= 4), "scale registers of the plain loop");
    if constexpr (NP == 2) {
        if constexpr (A_KC) {
#pragma unroll
            for (int i = 0; i < A_V4; ++i) scA_kc[i] = scale_of(g.a_amax, m0 + (tid + i * 256) / (KB / 4), g.M);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) scA_mc[e] = scale_of(g.a_amax, m0 + 4 * (tid % (BM / 4)) + e, g.M);
        }
        if constexpr (B_KC) {
#pragma unroll
            for (int i = 0; i < B_V4; ++i) scB_kc[i] = scale_of(g.b_amax, n0 + (tid + i * 256) / (KB / 4), g.N);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) scB_mc[e] = scale_of(g.b_amax, n0 + 4 * (tid % (BN / 4)) + e, g.N);
        }
    }
    auto scaleA_of = [&](int i) -> f32x4 { if constexpr (A_KC) { const float s = scA_kc[NP == 2 ? i : 0]; return f32x4{s, s, s, s}; } else return scA_mc; };
    auto scaleB_of = [&](int i) -> f32x4 { if constexpr (B_KC) { const float s = scB_kc[NP == 2 ? i : 0]; return f32x4{s, s, s, s}; } else return scB_mc; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

    f32x4 ra[PF][A_V4], rb[PF][B_V4];        // PF register sets, always indexed statically
    int kload[PF];
    TileLoader<A_V4> LA;
    TileLoader<B_V4> LB;
    loader_init<A_KC, BM, A_V4, KB>(LA, A, g.lda, m0, g.M, g.K, tid, vecA);
    loader_init<B_KC, BN, B_V4, KB>(LB, B, g.ldb, n0, g.N, g.K, tid, vecB);

    // FAST path operand fetch: buffer loads.  The per-thread byte offsets are computed once; the K advance is a scalar
    // (soffset) -- no vector ALU work per K-step (the split already uses ~3.7 VALU slots per MFMA) -- and rows beyond the
    // matrix end read as zero through the descriptor's bounds check.
    const long long spanA = A_KC ? ((long long)(g.M - 1) * g.lda + g.K) : ((long long)(g.K - 1) * g.lda + g.M);
    const long long spanB = B_KC ? ((long long)(g.N - 1) * g.ldb + g.K) : ((long long)(g.K - 1) * g.ldb + g.N);
    const bool buf_ok = spanA < (1ll << 29) && spanB < (1ll << 29);           // 32-bit byte offsets
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(min(spanA, (1ll << 29)) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)(min(spanB, (1ll << 29)) * 4), 0x00020000);
    unsigned voffA[A_V4], voffB[B_V4];
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
        const int idx = tid + i * 256;
        voffA[i] = A_KC ? ((unsigned)(m0 + idx / (KB / 4)) * (unsigned)g.lda + 4u * (idx % (KB / 4))) * 4u
                        : ((unsigned)(idx / (BM / 4)) * (unsigned)g.lda + (unsigned)m0 + 4u * (idx % (BM / 4))) * 4u;
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
        const int idx = tid + i * 256;
        voffB[i] = B_KC ? ((unsigned)(n0 + idx / (KB / 4)) * (unsigned)g.ldb + 4u * (idx % (KB / 4))) * 4u
                        : ((unsigned)(idx / (BN / 4)) * (unsigned)g.ldb + (unsigned)n0 + 4u * (idx % (BN / 4))) * 4u;
    }
    const bool ragged = ((kend - kbeg) % KB) != 0 || !vecA || !vecB || !buf_ok ||
                        (!A_KC && (m0 + BM > g.M)) || (!B_KC && (n0 + BN > g.N));
    auto lstore = [&](auto set_tag, bool need_mask) {
        constexpr int S = decltype(set_tag)::value;
        if (need_mask) {
            loader_mask<A_KC, BM, A_V4, KB>(LA, ra[S], kload[S], kend, tid);
            loader_mask<B_KC, BN, B_V4, KB>(LB, rb[S], kload[S], kend, tid);
        }
        if constexpr (ASUM) if (do_asum && asum_mine()) {
#pragma unroll
            for (int i = 0; i < A_V4; ++i) asum4 += ra[S][i];
        }
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
            const int idx = tid + i * 256;
            const int off = A_KC ? kc_store_off<KB>(idx) : mc_store_off<BM>(idx);
            u32x2 H, M, L;
            split4<NP>(ra[S][i], scaleA_of(i), H, M, L);
            *reinterpret_cast<u32x2*>(ldsA + off) = H;
            *reinterpret_cast<u32x2*>(ldsA + A_PLANE + off) = M;
            if constexpr (NP == 3) *reinterpret_cast<u32x2*>(ldsA + 2 * A_PLANE + off) = L;
        }
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
            const int idx = tid + i * 256;
            const int off = B_KC ? kc_store_off<KB>(idx) : mc_store_off<BN>(idx);
            u32x2 H, M, L;
            split4<NP>(rb[S][i], scaleB_of(i), H, M, L);
            *reinterpret_cast<u32x2*>(ldsB + off) = H;
            *reinterpret_cast<u32x2*>(ldsB + B_PLANE + off) = M;
            if constexpr (NP == 3) *reinterpret_cast<u32x2*>(ldsB + 2 * B_PLANE + off) = L;
        }
    };
    // per-lane offsets of the fragment reads (bytes inside a plane): see the image layouts above
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    int a_off[TM][NCH], b_off[TN][NCH];          // [tile][16-k chunk]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            a_off[i][c] = A_KC ? kc_read_off<KB>(wm + 32 * i + r, 2 * c + h) : mc_read_off<BM>(16 * c + 8 * h + tq, wm + 32 * i, tg, tp);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            b_off[j][c] = B_KC ? kc_read_off<KB>(wn + 32 * j + r, 2 * c + h) : mc_read_off<BN>(16 * c + 8 * h + tq, wn + 32 * j, tg, tp);
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    auto frag = [&](const char* plane, int off, bool kc, int rs) -> bf16x8 {
        if (kc) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(plane + off);
            return __builtin_bit_cast(bf16x8, v);
        } else {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off + 4 * rs));   // k + 4: same swizzle key
            const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, v);
        }
    };
    auto products = [&](f32x16& c, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP]) {
        if constexpr (NP == 2) {               // f16 planes (h, l): smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), c, 0, 0, 0);
            return;
        }
        if constexpr (NP == 3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    };
    auto compute = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bf16x8 fa[TM][NP], fb[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) fa[i][p] = frag(ldsA + p * A_PLANE, a_off[i][c], A_KC, A_RS);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) fb[j][p] = frag(ldsB + p * B_PLANE, b_off[j][c], B_KC, B_RS);
            // smallest terms first
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) products(acc[i][j], fa[i], fb[j]);
        }
    };
    auto mainloop = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        auto gload = [&](auto set_tag, int k0) {
            constexpr int S = decltype(set_tag)::value;
            kload[S] = k0;
            if constexpr (FAST) {
                const unsigned soA = (A_KC ? (unsigned)k0 : (unsigned)k0 * (unsigned)g.lda) * 4u;
                const unsigned soB = (B_KC ? (unsigned)k0 : (unsigned)k0 * (unsigned)g.ldb) * 4u;
#pragma unroll
                for (int i = 0; i < A_V4; ++i)
                    ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, voffA[i], soA, 0));
#pragma unroll
                for (int i = 0; i < B_V4; ++i)
                    rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, voffB[i], soB, 0));
            } else {
                loader_fetch<A_KC, BM, A_V4, KB, FAST>(LA, ra[S], g.lda, k0, g.K, g.M, m0, tid, vecA);
                loader_fetch<B_KC, BN, B_V4, KB, FAST>(LB, rb[S], g.ldb, k0, g.K, g.N, n0, tid, vecB);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, PF - 1>;
        if constexpr (PF == 1) {
            gload(S0{}, kbeg);
            lstore(S0{}, !FAST);
            __syncthreads();
            for (int k0 = kbeg; k0 < kend; k0 += KB) {
                const bool more = (k0 + KB) < kend;
#if EKAID_SPLIT_ABLATE == 1          // timing-only ablations (results are wrong): 1 = no global loads after the first tile
                (void)more;
                compute();
                __syncthreads();
                if (more) { lstore(S0{}, !FAST); __syncthreads(); }
#elif EKAID_SPLIT_ABLATE == 3        // 3 = no conversion / LDS stores after the first tile
                if (more) gload(S0{}, k0 + KB);
                compute();
                __syncthreads();
#elif EKAID_SPLIT_ABLATE == 4        // 4 = no MFMA phase
                if (more) gload(S0{}, k0 + KB);
                __syncthreads();
                if (more) { lstore(S0{}, !FAST); __syncthreads(); }
#elif EKAID_SPLIT_ABLATE == 5        // 5 = MFMA phase only
                compute();
#else
                if (more) gload(S0{}, k0 + KB);
                compute();
                __syncthreads();
                if (more) { lstore(S0{}, !FAST); __syncthreads(); }
#endif
            }
        } else {
            // two K-steps in flight; the loop is unrolled by two so that both register sets are indexed statically.
            // The loads are issued UNCONDITIONALLY (addresses are clamped into the matrix, a step beyond kend is
            // fetched and dropped): a load under a branch makes the compiler wait vmcnt(0) at the next LDS store,
            // which would drain the younger set as well and undo the second step in flight.
            gload(S0{}, kbeg);
            gload(S1{}, kbeg + KB);
            lstore(S0{}, !FAST);
            __syncthreads();
            for (int k0 = kbeg; k0 < kend; k0 += 2 * KB) {
                gload(S0{}, k0 + 2 * KB);                             // set 1 still holds step k0 + KB
                compute();
                __syncthreads();
                if (k0 + KB >= kend) break;
                lstore(S1{}, !FAST);
                __syncthreads();
                gload(S1{}, k0 + 3 * KB);
                compute();
                __syncthreads();
                if (k0 + 2 * KB < kend) { lstore(S0{}, !FAST); __syncthreads(); }
            }
        }
    };
    // ---- software-pipelined K loop (aligned, unragged problems; KB = 32): the LDS is used as TWO 16-deep stages.  In
    // stage s every wave multiplies out stage buffer s%2 while it converts the raw registers of stage s+1 into the other
    // buffer and has the loads of stage s+2 / s+3 in flight: one barrier per 16 k, and every wave carries MFMA and
    // vector-ALU work at the same time (with separate load / convert / multiply phases the co-resident workgroups of
    // a CU run their phases in step: measured 48 % MFMA busy + 35 % VALU busy with next to no overlap).
    auto pipelined = [&]() {
        constexpr int KS = 16;
        constexpr int PA = BM * KS / 4 / 256, PB = BN * KS / 4 / 256;       // float4 per thread per stage
        constexpr int SA_RS = A_KC ? 2 * KS : 2 * BM, SB_RS = B_KC ? 2 * KS : 2 * BN;
        constexpr int SA_PLANE = (A_KC ? BM : KS) * SA_RS, SB_PLANE = (B_KC ? BN : KS) * SB_RS;
        constexpr int STAGE_BYTES = NP * (SA_PLANE + SB_PLANE);
        unsigned vA[PA], vB[PB];
        int stA[PA], stB[PB];                 // LDS store offsets inside a plane
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int idx = tid + i * 256;
            if (A_KC) {
                const int row = idx >> 2, c4 = idx & 3;
                vA[i] = ((unsigned)(m0 + row) * (unsigned)g.lda + 4u * c4) * 4u;
                stA[i] = row * 32 + 16 * ((c4 >> 1) ^ ((row >> 3) & 1)) + 8 * (c4 & 1);
            } else {
                vA[i] = ((unsigned)(idx / (BM / 4)) * (unsigned)g.lda + (unsigned)m0 + 4u * (idx % (BM / 4))) * 4u;
                stA[i] = mc_store_off<BM>(idx);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int idx = tid + i * 256;
            if (B_KC) {
                const int row = idx >> 2, c4 = idx & 3;
                vB[i] = ((unsigned)(n0 + row) * (unsigned)g.ldb + 4u * c4) * 4u;
                stB[i] = row * 32 + 16 * ((c4 >> 1) ^ ((row >> 3) & 1)) + 8 * (c4 & 1);
            } else {
                vB[i] = ((unsigned)(idx / (BN / 4)) * (unsigned)g.ldb + (unsigned)n0 + 4u * (idx % (BN / 4))) * 4u;
                stB[i] = mc_store_off<BN>(idx);
            }
        }
        // fragment read offsets inside a stage plane: k-contiguous rows are 32 B (two 16-B slots, key = (row >> 3) & 1)
        int fA[TM], fB[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fA[i] = A_KC ? (wm + 32 * i + r) * 32 + 16 * (h ^ (((wm + 32 * i + r) >> 3) & 1)) : mc_read_off<BM>(8 * h + tq, wm + 32 * i, tg, tp);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fB[j] = B_KC ? (wn + 32 * j + r) * 32 + 16 * (h ^ (((wn + 32 * j + r) >> 3) & 1)) : mc_read_off<BN>(8 * h + tq, wn + 32 * j, tg, tp);

        f32x4 pa[2][PA], pb[2][PB];
        auto gload16 = [&](auto set_tag, int stage) {
            constexpr int S = decltype(set_tag)::value;
            const unsigned k = (unsigned)(kbeg + KS * stage);
            const unsigned soA = (A_KC ? k : k * (unsigned)g.lda) * 4u, soB = (B_KC ? k : k * (unsigned)g.ldb) * 4u;
#pragma unroll
            for (int i = 0; i < PA; ++i) pa[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, vA[i], soA, 0));
#pragma unroll
            for (int i = 0; i < PB; ++i) pb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, vB[i], soB, 0));
        };
        // scales of the 16-deep stages: a k-contiguous float4 lies in row idx >> 2
        f32x4 psA[A_KC ? PA : 1], psB[B_KC ? PB : 1];
        if constexpr (NP == 2 && A_KC) {
#pragma unroll
            for (int i = 0; i < PA; ++i) { const float s = scale_of(g.a_amax, m0 + ((tid + i * 256) >> 2), g.M); psA[i] = f32x4{s, s, s, s}; }
        }
        if constexpr (NP == 2 && B_KC) {
#pragma unroll
            for (int i = 0; i < PB; ++i) { const float s = scale_of(g.b_amax, n0 + ((tid + i * 256) >> 2), g.N); psB[i] = f32x4{s, s, s, s}; }
        }
        auto pscaleA = [&](int c) -> f32x4 { if constexpr (A_KC) return psA[NP == 2 ? c : 0]; else return scA_mc; };
        auto pscaleB = [&](int c) -> f32x4 { if constexpr (B_KC) return psB[NP == 2 ? c : 0]; else return scB_mc; };
        auto cvt1 = [&](const f32x4& v, const f32x4& s, char* base, int plane, int off) {
            u32x2 H, M, L;
            split4<NP>(v, s, H, M, L);
            *reinterpret_cast<u32x2*>(base + off) = H;
            *reinterpret_cast<u32x2*>(base + plane + off) = M;
            if constexpr (NP == 3) *reinterpret_cast<u32x2*>(base + 2 * plane + off) = L;
        };
        bool asum_cur = false;                    // is the 32-deep K-step being converted this workgroup's turn (A row sums)?
        // one stage: multiply out buffer `cur`, convert register set S into buffer `cur ^ 1`
        auto stage = [&](auto set_tag, int cur, bool convert) {
            constexpr int S = decltype(set_tag)::value;
            char* const cA = ldsA + cur * STAGE_BYTES;
            char* const cB = cA + NP * SA_PLANE;
            char* const nA = ldsA + (cur ^ 1) * STAGE_BYTES;
            char* const nB = nA + NP * SA_PLANE;
            // (the round-robin of the A row sums counts 32-deep K-steps, like the plain loop a ragged workgroup of the same
            // tile row runs: stages 2u and 2u + 1 share a turn)
            const bool mine = ASUM && convert && asum_cur;
            bf16x8 fa[TM][NP], fb[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) fa[i][p] = frag(cA + p * SA_PLANE, fA[i], A_KC, SA_RS);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) fb[j][p] = frag(cB + p * SB_PLANE, fB[j], B_KC, SB_RS);
            constexpr int NCV = PA + PB;                        // conversions to spread over the TM*TN product groups
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    products(acc[i][j], fa[i], fb[j]);
                    if (convert) {
                        const int grp = i * TN + j;
#pragma unroll
                        for (int c = 0; c < NCV; ++c) {
                            if (c * (TM * TN) / NCV != grp) continue;
                            if (c < PA) {
                                if constexpr (ASUM) if (mine) asum4 += pa[S][c];
                                cvt1(pa[S][c], pscaleA(c), nA, SA_PLANE, stA[c]);
                            } else cvt1(pb[S][c - PA], pscaleB(c - PA), nB, SB_PLANE, stB[c - PA]);
                        }
                    }
                }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        const int nst = (kend - kbeg) / KS;        // even: (kend - kbeg) % 32 == 0 on this path
        gload16(S0{}, 0);
        gload16(S1{}, 1);
        if constexpr (ASUM) asum_cur = do_asum && asum_mine();             // stage 0 (and 1)
#pragma unroll
        for (int c = 0; c < PA; ++c) {
            if constexpr (ASUM) if (asum_cur) asum4 += pa[0][c];
            cvt1(pa[0][c], pscaleA(c), ldsA, SA_PLANE, stA[c]);
        }
#pragma unroll
        for (int c = 0; c < PB; ++c) cvt1(pb[0][c], pscaleB(c), ldsA + NP * SA_PLANE, SB_PLANE, stB[c]);
        gload16(S0{}, 2);
        __syncthreads();
        for (int s0 = 0; s0 < nst; s0 += 2) {
#if EKAID_SPLIT_ABLATE == 6            // timing-only: no conversion / LDS stores in the stages
            stage(S1{}, 0, false);
            gload16(S1{}, s0 + 3);
            __syncthreads();
            stage(S0{}, 1, false);
            gload16(S0{}, s0 + 4);
            __syncthreads();
#elif EKAID_SPLIT_ABLATE == 7          // timing-only: no barriers
            stage(S1{}, 0, true);
            gload16(S1{}, s0 + 3);
            stage(S0{}, 1, s0 + 2 < nst);
            gload16(S0{}, s0 + 4);
#elif EKAID_SPLIT_ABLATE == 8          // timing-only: no global loads in the loop
            stage(S1{}, 0, true);
            __syncthreads();
            stage(S0{}, 1, s0 + 2 < nst);
            __syncthreads();
#else
            stage(S1{}, 0, true);                  // buffer 0 = stage s0; registers set 1 = stage s0+1 -> buffer 1
            if constexpr (ASUM) asum_cur = do_asum && asum_mine();          // stages s0+2, s0+3: the next 32-deep K-step
            gload16(S1{}, s0 + 3);
            __syncthreads();
            stage(S0{}, 1, s0 + 2 < nst);          // buffer 1 = stage s0+1; set 0 = stage s0+2 -> buffer 0
            gload16(S0{}, s0 + 4);
            __syncthreads();
#endif
        }
    };
    if (kbeg < kend) {
        if (ragged) mainloop(std::false_type{});
        else if constexpr (PIPE) pipelined();
        else mainloop(std::true_type{});
    }
    const int* sh_tab = nullptr;
    if constexpr (NP == 2) {
        // the shifts of the tile's rows and columns, for write_tiles, through LDS (free: the last stage has been multiplied out).  Kept
        // behind the A-row-sum scratch of the weight-gradient form (GR * BM floats from the start of the LDS)
        int* const sh = reinterpret_cast<int*>(lds) + (ASUM ? (256 / (BM / 4)) * BM : 0);
        __syncthreads();
        for (int t = tid; t < BM + BN; t += 256) {
            const bool isA = t < BM;
            const int idx = isA ? m0 + t : n0 + (t - BM);
            sh[t] = (idx < (isA ? g.M : g.N)) ? f16_shift((isA ? g.a_amax : g.b_amax)[idx]) : 0;
        }
        sh_tab = sh;
        __syncthreads();
    }
    if constexpr (ASUM) if (do_asum) {
        // the 256 / (BM / 4) threads that staged the same four rows are added up through LDS in thread order (deterministic)
        constexpr int GR = 256 / (BM / 4);
        __syncthreads();                                  // the last stage has been multiplied out: the LDS is free
        *reinterpret_cast<f32x4*>(&lds[(tid / (BM / 4)) * BM + 4 * (tid % (BM / 4))]) = asum4;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < GR; ++j) sum += lds[j * BM + tid];
            if (g.splitk > 1) g.asum_ws[((long long)z * asum_period + tcol) * g.M + m0 + tid] = sum;
            else g.asum[m0 + tid] += sum;
        }
    }
    write_tiles<TM, TN, NP == 2>(g, acc, m0, n0, wm, wn, r, h, z, sh_tab);
}

template <bool A_KC, bool B_KC, int TM, int TN, int KB, int PF, bool PIPE, int NP = 3, bool ASUM = false>
__global__ __launch_bounds__(256, (TM * TN >= 4 ? EKAID_SPLIT_OCC_BIG : (TM * TN == 2 ? 2 : EKAID_SPLIT_OCC_SMALL))) void gemm_split_kernel(GemmArgs g) {
    gemm_split_body<A_KC, B_KC, TM, TN, KB, PF, PIPE, NP, ASUM>(g, blockIdx.x, blockIdx.z);
}

// TWO independent products of the same operand layout in one launch (64x64 tiles): workgroups [0, tiles of g0) work on g0,
// the rest on g1; grid.z = the larger of the two K splits, a workgroup beyond its own problem's split count leaves at once.
// For the recurrent steps, where two small GEMMs of one step have no dependence on each other (ekaid_gemm_slabs2).
// Also the launch behind ekaid_gemm_pair_f32 (two full products with their own epilogues, e.g. the query / key projections of a
// graph-attention layer: 2 x 1152 tiles = three whole waves of resident workgroups instead of two launches of 1.5 waves each).
// ASUM: the A row sums of the weight-gradient form (either problem may carry one); PIPE: the K-loop arm of the 128x128 tile.
template <bool A_KC, bool B_KC, bool BIG, int NP = 3, bool ASUM = false, bool PIPE = EKAID_SPLIT_PIPE_BIG>
__global__ __launch_bounds__(256, BIG ? EKAID_SPLIT_OCC_BIG : EKAID_SPLIT_OCC_SMALL) void gemm_split_pair_kernel(GemmArgs g0, GemmArgs g1) {
    const int n0 = g0.tiles_m * g0.tiles_n;
    const bool first = (int)blockIdx.x < n0;
    if constexpr (NP == 2) {
        // two inlined copies of the tile body, each reading ITS problem straight from the kernel arguments: with one body working on
        // `first ? g0 : g1` every field is a select between two argument blocks held in scalar registers, and the two-term kernels -- four
        // more pointers, the scale tables -- ran out of them (100 SGPRs, then 23-49 spilled VGPRs and ~1500 scratch instructions in the
        // epilogue: the pairs ran SLOWER than two solo launches)
        if (first) {
            if ((int)blockIdx.z >= g0.splitk) return;
            if constexpr (BIG) gemm_split_body<A_KC, B_KC, 2, 2, 32, 1, PIPE, NP, ASUM>(g0, blockIdx.x, blockIdx.z);
            else gemm_split_body<A_KC, B_KC, 1, 1, EKAID_SPLIT_KB_SMALL, EKAID_SPLIT_PF_SMALL, EKAID_SPLIT_PIPE_SMALL, NP, ASUM>(g0, blockIdx.x, blockIdx.z);
        } else {
            if ((int)blockIdx.z >= g1.splitk) return;
            if constexpr (BIG) gemm_split_body<A_KC, B_KC, 2, 2, 32, 1, PIPE, NP, ASUM>(g1, blockIdx.x - n0, blockIdx.z);
            else gemm_split_body<A_KC, B_KC, 1, 1, EKAID_SPLIT_KB_SMALL, EKAID_SPLIT_PF_SMALL, EKAID_SPLIT_PIPE_SMALL, NP, ASUM>(g1, blockIdx.x - n0, blockIdx.z);
        }
        return;
    }
    const GemmArgs& g = first ? g0 : g1;
    if ((int)blockIdx.z >= g.splitk) return;
    const int bx = first ? blockIdx.x : blockIdx.x - n0;
    if constexpr (BIG) gemm_split_body<A_KC, B_KC, 2, 2, 32, 1, PIPE, NP, ASUM>(g, bx, blockIdx.z);
    else gemm_split_body<A_KC, B_KC, 1, 1, EKAID_SPLIT_KB_SMALL, EKAID_SPLIT_PF_SMALL, EKAID_SPLIT_PIPE_SMALL, NP, ASUM>(g, bx, blockIdx.z);
}

#ifndef EKAID_GEMM_X2_TU      // (non-template kernels: defined once, in the main translation unit)
// ---- tall-skinny wgrad: C[M][N] = A^T . B with M * N <= 256 outputs and a very long contraction (the label-bias and
// pair-geometry layers: 1 x 11 and 4 x 64 outputs over K = R * N * N = 663 552 rows at B = 256).  As a tiled GEMM this is one
// 64x64 tile cut into ~770 K slabs with scalar operand loads (M, N not multiples of 4): 113-146 us for 30-180 MB.  Here:
// thread (row lane rl, output o) of a workgroup walks its K chunk with stride RL = 256 / NO (NO = outputs rounded up to a
// power of two), so a wave reads RL consecutive rows of both operands (contiguous memory); lanes are combined through LDS,
// workgroup partials by a fixed-order finish kernel (deterministic, no atomics).
__global__ __launch_bounds__(256) void tn_skinny_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        int M, int N, int K, int rows_per_wg, int NO, float* __restrict__ partial) {
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    const int o = tid & (NO - 1), rl = tid / NO, RL = 256 / NO;
    const int MN = M * N;
    const int m = o < MN ? o / N : 0, n = o < MN ? o % N : 0;
    const int k0 = blockIdx.x * rows_per_wg, k1 = min(K, k0 + rows_per_wg);
    const float* ap = A + (long long)(k0 + rl) * lda + m;
    const float* bp = B + (long long)(k0 + rl) * ldb + n;
    const long long sa = (long long)RL * lda, sb = (long long)RL * ldb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = k0 + rl;
    for (; k + 3 * RL < k1; k += 4 * RL) {          // four independent row pairs in flight
        const float a0 = ap[0], a1 = ap[sa], a2 = ap[2 * sa], a3 = ap[3 * sa];
        const float b0 = bp[0], b1 = bp[sb], b2 = bp[2 * sb], b3 = bp[3 * sb];
        s0 = fmaf(a0, b0, s0); s1 = fmaf(a1, b1, s1); s2 = fmaf(a2, b2, s2); s3 = fmaf(a3, b3, s3);
        ap += 4 * sa; bp += 4 * sb;
    }
    for (; k < k1; k += RL) { s0 = fmaf(ap[0], bp[0], s0); ap += sa; bp += sb; }
    sm[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < NO) {
        float s = 0.f;
        for (int q = 0; q < RL; ++q) s += sm[q * NO + tid];
        if (tid < MN) partial[(long long)blockIdx.x * MN + tid] = s;
    }
}
// M <= 4 with a wide N: thread (row lane, n) keeps all M outputs of column n -- every B word is loaded once per workgroup
// (the one-output-per-thread mapping above makes the M waves of a workgroup fetch the same B row)
template <int MT>
__global__ __launch_bounds__(256) void tn_skinny_cols_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                             int M, int N, int K, int rows_per_wg, int NO, float* __restrict__ partial) {
    __shared__ float sm[MT][256];
    const int tid = threadIdx.x;
    const int n = tid & (NO - 1), rl = tid / NO, RL = 256 / NO;
    const int nc = n < N ? n : 0;
    const int k0 = blockIdx.x * rows_per_wg, k1 = min(K, k0 + rows_per_wg);
    const float* ap = A + (long long)(k0 + rl) * lda;
    const float* bp = B + (long long)(k0 + rl) * ldb + nc;
    const long long sa = (long long)RL * lda, sb = (long long)RL * ldb;
    float s[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) s[m][0] = s[m][1] = 0.f;
    int k = k0 + rl;
    for (; k + 3 * RL < k1; k += 4 * RL) {
        float b[4], a[4][MT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            b[u] = bp[u * sb];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[u][m] = m < M ? ap[u * sa + m] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) s[m][u & 1] = fmaf(a[u][m], b[u], s[m][u & 1]);
        ap += 4 * sa; bp += 4 * sb;
    }
    for (; k < k1; k += RL) {
        const float b0 = bp[0];
#pragma unroll
        for (int m = 0; m < MT; ++m) s[m][0] = fmaf(m < M ? ap[m] : 0.f, b0, s[m][0]);
        ap += sa; bp += sb;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) sm[m][tid] = s[m][0] + s[m][1];
    __syncthreads();
    if (tid < NO && tid < N) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < M) {
                float t = 0.f;
                for (int q = 0; q < RL; ++q) t += sm[m][q * NO + tid];
                partial[(long long)blockIdx.x * M * N + m * N + tid] = t;
            }
        }
    }
}
__global__ __launch_bounds__(256) void tn_skinny_finish_kernel(const float* __restrict__ partial, int G, int MN, int N, float* C, int ldc,
                                                               float alpha, int accumulate) {
    __shared__ float sm[256];
    const int o = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int gidx = tid; gidx < G; gidx += 256) s += partial[(long long)gidx * MN + o];
    sm[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) sm[tid] += sm[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        float* cp = C + (long long)(o / N) * ldc + o % N;
        *cp = alpha * sm[0] + (accumulate ? *cp : 0.f);
    }
}

// the K-split partials of the A row sums (GemmArgs::asum_ws) are finished by the first threads of the reduce grid, slab order
__device__ __forceinline__ void asum_finish(const float* asum_ws, float* asum, int M, int nparts, long long gtid) {
    if (asum == nullptr || gtid >= M) return;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nparts; k += 8) {             // eight independent loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = asum_ws[(long long)(k + u) * M + gtid];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nparts; ++k) s += asum_ws[(long long)k * M + gtid];
    asum[gtid] += s;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, float* C, int M, int N, int ldc,
                                                            int splitk, Epi e, const float* asum_ws, float* asum, int asum_parts) {
    const long long total = (long long)M * N;
    asum_finish(asum_ws, asum, M, asum_parts, (long long)blockIdx.x * blockDim.x + threadIdx.x);
    const float alpha = e.alpha_dev ? e.alpha * e.alpha_dev[0] : e.alpha;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += ws[(long long)k * total + i];
        float* cp = C + (long long)m * ldc + n;
        const float cold = e.beta ? *cp : 0.f;
        *cp = epilogue(s, cold, epi_pre(e, m, n), e.post ? e.post[(long long)m * e.ldpost + n] : 0.f, epi_seed(e), m, n, e, alpha);
    }
}

// 16-byte variant (N and ldc multiples of 4, 16-byte aligned C): one float4 of the output per thread, the S slab
// loads of a thread are independent (all in flight together), 32-bit index math only
template <int S_UNROLL>
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N4,
                                                                int ldc, int splitk, Epi e, const float* asum_ws, float* asum, int asum_parts) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    asum_finish(asum_ws, asum, M, asum_parts, idx);        // M <= M * N4: the grid covers it
    if (idx >= M * N4) return;
    const int m = idx / N4, n = (idx - m * N4) * 4;
    const long long total4 = (long long)M * N4;
    const f32x4* w = reinterpret_cast<const f32x4*>(ws) + idx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + S_UNROLL <= splitk; k += S_UNROLL) {
        f32x4 v[S_UNROLL];
#pragma unroll
        for (int u = 0; u < S_UNROLL; ++u) v[u] = w[(long long)(k + u) * total4];
#pragma unroll
        for (int u = 0; u < S_UNROLL; ++u) s += v[u];
    }
    for (; k < splitk; ++k) s += w[(long long)k * total4];
    const float alpha = e.alpha_dev ? e.alpha * e.alpha_dev[0] : e.alpha;
    f32x4* cp = reinterpret_cast<f32x4*>(C + (long long)m * ldc + n);
    f32x4 cold = {0.f, 0.f, 0.f, 0.f};
    if (e.beta) cold = *cp;
    f32x4 o;
    f32x4 pv = {0.f, 0.f, 0.f, 0.f};
    if (e.post) {
        const float* pp = e.post + (long long)m * e.ldpost + n;
        if ((e.ldpost & 3) == 0 && (reinterpret_cast<uintptr_t>(e.post) & 15) == 0) pv = *reinterpret_cast<const f32x4*>(pp);
        else { pv[0] = pp[0]; pv[1] = pp[1]; pv[2] = pp[2]; pv[3] = pp[3]; }
    }
    const uint64_t seed = epi_seed(e);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = epilogue(s[c], cold[c], epi_pre(e, m, n + c), pv[c], seed, m, n + c, e, alpha);
    *cp = o;
}

#endif

template <bool A_KC, bool B_KC, int TM, int TN, int KB>
static int launch(const GemmArgs& a, int batch, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_ELEMS = A_KC ? BM * (KB + LDS_PAD) : KB * (BM + LDS_PAD);
    constexpr int B_ELEMS = B_KC ? BN * (KB + LDS_PAD) : KB * (BN + LDS_PAD);
    constexpr size_t lds = (A_ELEMS + B_ELEMS) * sizeof(float);
    auto kern = gemm_mfma_kernel<A_KC, B_KC, TM, TN, KB>;
    if (lds > 64 * 1024) {
        if (func_attr_once((const void*)kern)) {                                               // once per device and instantiation
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                set_error("gemm: cannot raise the dynamic LDS limit to %zu bytes", lds);
                return EKAID_ERR_LAUNCH;
            }
        }
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.splitk > 1 ? a.splitk : batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return EKAID_OK;
}

template <bool A_KC, bool B_KC, int TM, int TN, int KB, int PF, bool PIPE, int NP = 3, bool ASUM = false>
static int launch_split(const GemmArgs& a, int batch, hipStream_t st) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr size_t lds = NP * 2 * KB * (BM + BN);     // NP bf16 planes per operand, unpadded
    auto kern = gemm_split_kernel<A_KC, B_KC, TM, TN, KB, PF, PIPE, NP, ASUM>;
    if (lds > 64 * 1024) {
        if (func_attr_once((const void*)kern)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                set_error("gemm: cannot raise the dynamic LDS limit to %zu bytes", lds);
                return EKAID_ERR_LAUNCH;
            }
        }
    }
    dim3 grid(a.tiles_m * a.tiles_n, 1, a.splitk > 1 ? a.splitk : batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    return EKAID_OK;
}

// arithmetic of the MFMA kernels: 0 = v_mfma_f32_32x32x2_f32, 1 = exact 3-way bf16 split (6 x v_mfma_f32_32x32x16_bf16);
// kept in the per-device state (api.hip: gemm_mode / set_gemm_mode)

static thread_local int t_plan_slab = 0;      // set while a slab-mode product is being planned (fill_args)

static int plan_splitk(int tiles, int K, int batch, long long MN, int BK, bool big_tiles) {
    if (batch > 1) return 1;
    static const char* tgt = getenv("EKAID_GEMM_SPLIT_TARGET");   // tuning aids
    static const char* tgs = getenv("EKAID_GEMM_SPLIT_TARGET_SMALL");
    const int target = (!big_tiles && tgs) ? atoi(tgs) : (tgt ? atoi(tgt) : 768);     // ~3 workgroups per CU (the register budget admits 3 per SIMD)
    if (tiles * 4 >= target * 3) return 1;
    // 64x64 tiles, one workgroup per CU already: a 2-way split + reduce kernel measured no faster (256x6656x512: 23 vs 24 us).
    // (128x128 tiles are different: 256 tiles leave two of the three workgroup slots of every CU empty for the whole K
    // range -- wgrad 1024x4096x18432: 156-162 TF unsplit, 199 TF as 3 x 256 workgroups + reduce)
    // (except a long contraction: the weight-gradient products that plan() moves to this tile at small batch, 256 tiles x K >= 1536)
    if (tiles >= 256 && gemm_mode() >= 1 && !big_tiles && K < 1536) return 1;
    // as many slabs as still fit ONE wave of resident workgroups: rounding up (144 tiles -> 6 slabs = 864 workgroups on 768
    // slots) left a second, nearly empty wave that doubled the time of the 2304-row products (61 -> 44 us at 32 pairs / GPU)
    int s = target / tiles;
    if (s < 1) s = 1;
    const int ksteps = cdiv(K, BK);
    static const char* mk = getenv("EKAID_GEMM_MIN_KSTEPS");     // tuning aid
    const int min_ksteps = mk ? atoi(mk) : (128 / BK);     // a split shorter than this is all prologue/epilogue + slab traffic
    s = s > ksteps / min_ksteps ? ksteps / min_ksteps : s;
    while (s > 1 && (long long)s * MN * 4 > (1ll << 30)) --s;   // cap the slab workspace at 1 GiB
    return s < 1 ? 1 : s;
}

static void plan(const ekaid_gemm_t* d, int& TMN, int& splitk, int& tiles_m, int& tiles_n) {
    const int batch = d->batch1 * d->batch2;
    // 128x128 tiles (split along K when the grid is small) for problems with many rows; 64x64 tiles
    // for the small-M recurrent GEMMs and narrow outputs, where slab traffic would rival the math.
    (void)batch;
    static const char* br = getenv("EKAID_GEMM_BIG_ROWS");        // tuning aid
    const int big_rows = br ? atoi(br) : 512;
    TMN = (d->M >= big_rows && d->N >= 128) ? 2 : 1;
    // tall outputs with few column tiles (forward/dgrad of the 1024-wide layers, M = 2*B*N rows): 128x64 tiles run
    // 4-5 workgroups per CU and quantise better than 128x128 (measured 94 -> 104 TF NT, 114 -> 122 TF NN at N = 1024)
    // (f32 MFMA kernel only: the bf16-split kernel runs 3 workgroups of 128x128 per CU and is faster with them)
    if (TMN == 2 && batch == 1 && d->M >= 4096 && d->N <= 2048 && gemm_mode() == 0) TMN = 3;
    // bf16-split arithmetic, tall outputs with ONE 128-wide column tile or a 128-deep contraction (the 128-unit attention
    // layers, M = 9216): 72 tiles of 128x128 cannot fill the chip; measured 35.0 -> 28.5 us (nt 9216x128x1024 as 128x64
    // tiles) and 35.1 -> 28.7 us (nn 9216x1024x128 as 64x64 tiles)
    if (TMN == 2 && batch == 1 && gemm_mode() >= 1 && d->M >= 4096 && !d->transA) {
        if (d->transB && d->N <= 128) TMN = 3;
        else if (!d->transB && d->K <= 128) TMN = 1;
    }
    {
        // few 128x128 tiles but enough 64x64 ones to fill the chip WITHOUT a K split (the 2304- / 4608-row products of a 32- /
        // 64-pair step: 144 tiles x 5 slabs + reduce kernel vs 576 unsplit tiles), short contractions only: same box, hipGraph
        // replay, 9.55 -> 9.50 ms at 32 pairs, 12.68 -> 12.43 at 64, 19.40 -> 19.33 at 128; with K up to 4096 it loses
        // (9.57 -> 9.71).  EKAID_GEMM_SMALL_UNSPLIT=<max K> overrides (0 = off).
        static const char* su = getenv("EKAID_GEMM_SMALL_UNSPLIT");
        const int su_k = su ? atoi(su) : 1024;
        if (TMN == 2 && batch == 1 && gemm_mode() >= 1) {
            const long long t128 = (long long)cdiv(d->M, 128) * cdiv(d->N, 128), t64 = (long long)cdiv(d->M, 64) * cdiv(d->N, 64);
            if (t128 * 2 < 768 && t64 >= 512 && d->K <= su_k) TMN = 1;     // (under half a wave of 128x128 tiles: at 416 tiles -- 6656 rows, the
                                                                             // real-data shape -- the big tile wins again: 103-106 vs 111-115 us)
        }
    }
    {
        // a weight-gradient product whose 128x128 grid needs so many K slabs to fill the chip that each slab is only a few K-steps
        // deep (1024x1024x2304 at 32 pairs: 64 tiles x 12 slabs of 192 -- all prologue, epilogue and slab traffic) runs on 64x64
        // tiles with a 3-way split instead (256 tiles x 3 slabs of 768).  Same box, hipGraph replay: 9.35 -> 9.05 ms at 32 pairs,
        // 12.25 -> 12.18 at 64, nothing applies at 256.  EKAID_GEMM_WGRAD_SMALL=<min slab depth> overrides (0 = off).
        static const char* wsm = getenv("EKAID_GEMM_WGRAD_SMALL");
        const int min_depth = wsm ? atoi(wsm) : 512;
        if (TMN == 2 && batch == 1 && gemm_mode() >= 1 && d->transA) {
            const long long t128 = (long long)cdiv(d->M, 128) * cdiv(d->N, 128), t64 = (long long)cdiv(d->M, 64) * cdiv(d->N, 64);
            const long long slabs = t128 > 0 ? 768 / t128 : 1;
            if (slabs >= 2 && d->K / slabs < min_depth && t64 >= 192 && t64 <= 768) TMN = 1;
        }
    }
    {
        // slab-mode products (consumer-reduced, the recurrent steps) with 64 < M: tuning aid EKAID_GEMM_SLAB_TILE=3 (128x64) / 4
        static const char* st = getenv("EKAID_GEMM_SLAB_TILE");
        if (st && t_plan_slab && TMN == 1 && d->M > 64 && (st[0] == '3' || st[0] == '4')) TMN = st[0] - '0';
    }
    static const char* force = getenv("EKAID_GEMM_TILE");         // tuning aid: "1" (64x64), "2" (128x128), "3" (128x64), "4" (64x128)
    if (force && force[0] >= '1' && force[0] <= '4' && (force[0] < '3' || TMN == 2)) TMN = force[0] - '0';
    const int bm = TMN == 4 ? 64 : (TMN == 3 ? 128 : 64 * TMN);
    const int bn = TMN == 3 ? 64 : (TMN == 4 ? 128 : 64 * TMN);
    tiles_m = cdiv(d->M, bm);
    tiles_n = cdiv(d->N, bn);
    splitk = plan_splitk(tiles_m * tiles_n, d->K, batch, (long long)d->M * d->N, TMN >= 2 ? BK_BIG : (gemm_mode() >= 1 && TMN == 1 ? EKAID_SPLIT_KB_SMALL : BK_SMALL), TMN == 2);
}

// shapes of tn_skinny_kernel: (workgroups, outputs rounded up to a power of two), or false
static bool skinny_tn(const ekaid_gemm_t* d, int& G, int& NO) {
    const char* e = getenv("EKAID_GEMM_SKINNY");         // tuning aid / A-B tests: 0 = tiled kernels only
    if (e && e[0] == '0') return false;
    const long long MN = (long long)d->M * d->N;
    if (!d->transA || d->transB || d->batch1 * d->batch2 != 1 || MN > 256 || d->K < 8192 || d->bias || d->R || d->act != 0 ||
        d->drop_p > 0.f || d->alpha_dev || d->post)
        return false;
    NO = 16;
    if (d->M <= 4 && d->N >= 32) {               // tn_skinny_cols_kernel: NO = columns rounded up
        while (NO < d->N) NO <<= 1;
    } else {
        while (NO < MN) NO <<= 1;
    }
    G = d->K / ((NO >= 128 || (d->M <= 4 && d->N >= 32)) ? 128 : 512);     // few row lanes per workgroup: shorter chunks, more workgroups
    G = G > 4096 ? 4096 : (G < 1 ? 1 : G);
    return true;
}


// planner + the K range of one slab (empty trailing slabs dropped): what every entry point launches
static void plan_full(const ekaid_gemm_t* d, int& TMN, int& splitk, int& k_per_split, int& tiles_m, int& tiles_n) {
    plan(d, TMN, splitk, tiles_m, tiles_n);
    if (d->K == 0) splitk = 1;
    const int BK = (TMN == 3 || TMN == 4 ? BK_HALF : (TMN >= 2 ? BK_BIG : (gemm_mode() >= 1 && TMN == 1 ? EKAID_SPLIT_KB_SMALL : BK_SMALL)));
    k_per_split = cdiv(cdiv(d->K, BK), splitk) * BK;
    if (splitk > 1) {
        splitk = cdiv(d->K, k_per_split);
        if (splitk <= 1) { splitk = 1; k_per_split = cdiv(d->K, BK) * BK; }
    }
}

// GemmArgs of one product: tiling + K split from the planner, slab workspace checked.  slab_mode: raw partial products into
// `workspace` as [splitk][M][N] even when the planner does not split (ekaid_gemm_slabs).
static int fill_args(const ekaid_gemm_t* d, GemmArgs& a, int& TMN, void* workspace, size_t workspace_bytes, int slab_mode) {
    a.A = d->A; a.B = d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.sA1 = d->strideA1; a.sA2 = d->strideA2; a.sB1 = d->strideB1; a.sB2 = d->strideB2;
    a.sC1 = d->strideC1; a.sC2 = d->strideC2; a.nb2 = d->batch2;
    a.e.alpha = d->alpha; a.e.alpha_dev = d->alpha_dev; a.e.beta = d->accumulate; a.e.bias = d->bias;
    a.e.R = d->R; a.e.ldr = d->ldr; a.e.rdiv = d->rdiv < 1 ? 1 : d->rdiv; a.e.rowmask = d->rowmask; a.e.act = d->act;
    a.e.drop_p = d->drop_p; a.e.drop_inv = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f; a.e.drop_seed = d->drop_seed; a.e.drop_epoch = dropout_epoch_ptr(); a.e.n_cols = d->N;
    a.e.post = slab_mode ? nullptr : d->post; a.e.ldpost = d->ldpost;
    a.slab_mode = slab_mode;
    a.asum = nullptr; a.asum_ws = nullptr;
    a.a_amax = d->a_amax; a.b_amax = d->b_amax;
    t_plan_slab = slab_mode;
    plan_full(d, TMN, a.splitk, a.k_per_split, a.tiles_m, a.tiles_n);
    t_plan_slab = 0;
    a.ws = nullptr;
    if (a.splitk > 1 || slab_mode) {
        const size_t need = (size_t)a.splitk * d->M * d->N * sizeof(float);
        if (!workspace || workspace_bytes < need) {
            set_error("ekaid_gemm: workspace too small (%zu < %zu)", workspace_bytes, need);
            return EKAID_ERR_ARG;
        }
        a.ws = (float*)workspace;
    }
    return EKAID_OK;
}


// 128x128-tile arm (EKAID_GEMM_BIG = pipe | plain overrides): the software-pipelined loop (two 16-deep LDS stages) or the plain
// 32-deep loop.  Default: plain when both operands are k-contiguous (forward products: +4 % at K = 1024, +2.6 % at K = 4096),
// pipelined otherwise (+2 % on the weight-gradient form, equal on the input-gradient form).  A third arm on
// v_mfma_f32_16x16x32_bf16 was built, measured 3-8 % slower (its 198-213 registers allow 2 workgroups per CU only; counters:
// MFMA busy 0.39 at 2.14 GHz against 0.52 at 1.87 GHz) and moved to tools/experiments: profiles/r03_gemm_arms.txt.
static int big_arm(bool akc, bool bkc) {
    static const char* e = getenv("EKAID_GEMM_BIG");
    if (e) return (e[0] == 'p' && e[1] == 'l') ? 1 : 0;
    return (akc && bkc) ? 1 : 0;
}

// the split-arithmetic kernels of one plane count: tile shape TMN from the planner, K-loop arm per operand layout
template <int NP>
static int split_dispatch(const GemmArgs& a, int TMN, bool akc, bool bkc, int batch, hipStream_t st) {
#define EKAID_LAUNCH_SPLIT(AK, BKC) (TMN == 2 ? (big_arm(AK, BKC) == 1 ? launch_split<AK, BKC, 2, 2, 32, 1, false, NP>(a, batch, st) \
                                                  : launch_split<AK, BKC, 2, 2, 32, 1, EKAID_SPLIT_PIPE_BIG, NP>(a, batch, st)) \
                                     : TMN == 1 ? launch_split<AK, BKC, 1, 1, EKAID_SPLIT_KB_SMALL, EKAID_SPLIT_PF_SMALL, EKAID_SPLIT_PIPE_SMALL, NP>(a, batch, st) \
                                     : TMN == 3 ? launch_split<AK, BKC, 2, 1, 32, 1, false, NP>(a, batch, st) : launch_split<AK, BKC, 1, 2, 32, 1, false, NP>(a, batch, st))
    if (akc && bkc) return EKAID_LAUNCH_SPLIT(true, true);
    if (akc && !bkc) return EKAID_LAUNCH_SPLIT(true, false);
    if (a.asum != nullptr) {
        // weight-gradient form with the bias gradient riding on the A operand: the default loop arm of each tile shape only
        return TMN == 2 ? launch_split<false, false, 2, 2, 32, 1, EKAID_SPLIT_PIPE_BIG, NP, true>(a, batch, st)
             : TMN == 1 ? launch_split<false, false, 1, 1, EKAID_SPLIT_KB_SMALL, EKAID_SPLIT_PF_SMALL, false, NP, true>(a, batch, st)
             : TMN == 3 ? launch_split<false, false, 2, 1, 32, 1, false, NP, true>(a, batch, st)
                        : launch_split<false, false, 1, 2, 32, 1, false, NP, true>(a, batch, st);
    }
    return EKAID_LAUNCH_SPLIT(false, false);
#undef EKAID_LAUNCH_SPLIT
}
// (the reduced-precision instantiations live in their own translation unit, gemm_x2.hip, so that the two compile in parallel)
int split_dispatch_x2(const GemmArgs& a, int TMN, bool akc, bool bkc, int batch, hipStream_t st);
int pair_dispatch_x2(const GemmArgs& a0, const GemmArgs& a1, bool big, bool akc, bool bkc, hipStream_t st);

#ifndef EKAID_GEMM_X2_TU
// mode 2 (two-term f16 split) serves the products whose caller supplied bounds on both operands; every other product of the device
// keeps the exact 3-way bf16 split
static bool use_x2(const GemmArgs& a) { return gemm_mode() == 2 && a.a_amax != nullptr && a.b_amax != nullptr; }
// the product kernel of one planned GEMM (arithmetic per device state, tile shape TMN from the planner)
static int launch_product(const GemmArgs& a, int TMN, bool akc, bool bkc, int batch, hipStream_t st) {
    int rc;
    if (gemm_mode() >= 1 && TMN >= 1 && TMN <= 4) {
        const bool x2 = use_x2(a) && batch == 1;          // (the bound vectors have no batch strides)
        rc = x2 ? split_dispatch_x2(a, TMN, akc, bkc, batch, st) : split_dispatch<3>(a, TMN, akc, bkc, batch, st);
        count_launch(x2 ? EKAID_COUNT_GEMM_SPLIT2 : EKAID_COUNT_GEMM_SPLIT3);
    } else {
        count_launch(EKAID_COUNT_GEMM_F32);
#define EKAID_LAUNCH(AK, BKC) (TMN == 2 ? launch<AK, BKC, 2, 2, BK_BIG>(a, batch, st) : TMN == 1 ? launch<AK, BKC, 1, 1, BK_SMALL>(a, batch, st) \
                               : TMN == 3 ? launch<AK, BKC, 2, 1, BK_HALF>(a, batch, st) : launch<AK, BKC, 1, 2, BK_HALF>(a, batch, st))
        if (akc && bkc) rc = EKAID_LAUNCH(true, true);
        else if (akc && !bkc) rc = EKAID_LAUNCH(true, false);
        else rc = EKAID_LAUNCH(false, false);
#undef EKAID_LAUNCH
    }
    if (rc) return rc;
    return check_launch("gemm kernel");
}
#endif

template <bool AK, bool BKC, bool BIG, int NP = 3, bool ASUM = false, bool PIPE = EKAID_SPLIT_PIPE_BIG>
static int launch_pair(const GemmArgs& a0, const GemmArgs& a1, hipStream_t st) {
    constexpr int KB = BIG ? 32 : EKAID_SPLIT_KB_SMALL;
    constexpr size_t lds = NP * 2 * KB * (BIG ? 256 : 128);
    static_assert(lds <= 64 * 1024, "pair kernel: the tiles fit the default dynamic LDS limit");
    const int sk = a0.splitk > a1.splitk ? a0.splitk : a1.splitk;
    dim3 grid(a0.tiles_m * a0.tiles_n + a1.tiles_m * a1.tiles_n, 1, sk);
    hipLaunchKernelGGL((gemm_split_pair_kernel<AK, BKC, BIG, NP, ASUM, PIPE>), grid, dim3(256), lds, st, a0, a1);
    return check_launch("gemm_split_pair_kernel");
}
// (the 128x128 tile takes the K-loop arm a solo launch of the same layout takes -- big_arm -- so that a product computed inside a pair is
// bit-identical to the same product launched alone)
template <int NP>
static int pair_dispatch(const GemmArgs& a0, const GemmArgs& a1, bool big, bool akc, bool bkc, hipStream_t st) {
    const bool asum = a0.asum != nullptr || a1.asum != nullptr;
    if (asum) {
        if (akc || bkc) { set_error("gemm pair: A row sums need the weight-gradient layout"); return EKAID_ERR_UNSUPPORTED; }
        return big ? launch_pair<false, false, true, NP, true, EKAID_SPLIT_PIPE_BIG>(a0, a1, st)
                   : launch_pair<false, false, false, NP, true>(a0, a1, st);
    }
    if (big) {
        if (akc && bkc) return big_arm(true, true) == 1 ? launch_pair<true, true, true, NP, false, false>(a0, a1, st)
                                                        : launch_pair<true, true, true, NP, false, EKAID_SPLIT_PIPE_BIG>(a0, a1, st);
        if (akc && !bkc) return big_arm(true, false) == 1 ? launch_pair<true, false, true, NP, false, false>(a0, a1, st)
                                                          : launch_pair<true, false, true, NP, false, EKAID_SPLIT_PIPE_BIG>(a0, a1, st);
        return big_arm(false, false) == 1 ? launch_pair<false, false, true, NP, false, false>(a0, a1, st)
                                          : launch_pair<false, false, true, NP, false, EKAID_SPLIT_PIPE_BIG>(a0, a1, st);
    }
    return akc && bkc ? launch_pair<true, true, false, NP>(a0, a1, st)
         : akc && !bkc ? launch_pair<true, false, false, NP>(a0, a1, st) : launch_pair<false, false, false, NP>(a0, a1, st);
}
#if defined(EKAID_GEMM_X2_TU) && !defined(EKAID_ONE_KERNEL_TU)      // (EKAID_ONE_KERNEL_TU: tools' single-instantiation builds)
int split_dispatch_x2(const GemmArgs& a, int TMN, bool akc, bool bkc, int batch, hipStream_t st) { return split_dispatch<2>(a, TMN, akc, bkc, batch, st); }
int pair_dispatch_x2(const GemmArgs& a0, const GemmArgs& a1, bool big, bool akc, bool bkc, hipStream_t st) { return pair_dispatch<2>(a0, a1, big, akc, bkc, st); }
#endif
}  // namespace ekaid

#ifndef EKAID_GEMM_X2_TU
using namespace ekaid;

extern "C" int ekaid_set_gemm_mode(int mode) {
    if (mode < 0 || mode > 2) { set_error("ekaid_set_gemm_mode: mode must be 0 (f32 MFMA), 1 (exact bf16 3-way split) or 2 (two-term f16 split where operand bounds are given)"); return EKAID_ERR_ARG; }
    return set_gemm_mode(mode);
}
extern "C" int ekaid_get_gemm_mode(void) { return gemm_mode(); }

extern "C" size_t ekaid_gemm_workspace_bytes(const ekaid_gemm_t* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    int TMN, splitk, tm, tn;
    {
        int G, NO;
        if (skinny_tn(d, G, NO)) return (size_t)G * d->M * d->N * sizeof(float);
    }
    plan(d, TMN, splitk, tm, tn);
    // (+ the K-split partials of the A row sums, d->a_rowsum, behind the slabs)
    return splitk > 1 ? (size_t)splitk * d->M * ((size_t)d->N + (d->a_rowsum ? ASUM_SHARE : 0)) * sizeof(float) : 0;
}

extern "C" int ekaid_gemm_plan_query(const ekaid_gemm_t* d, int* tile, int* splitk) {
    if (!d || !tile || !splitk) { set_error("ekaid_gemm_plan_query: null pointer"); return EKAID_ERR_ARG; }
    *tile = 0; *splitk = 1;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return EKAID_OK;
    int G, NO;
    if (skinny_tn(d, G, NO)) { *splitk = G; return EKAID_OK; }       // tile 0: the tall-skinny reduction kernels
    int TMN, sk, kps, tm, tn;
    plan_full(d, TMN, sk, kps, tm, tn);
    *tile = TMN; *splitk = sk;
    return EKAID_OK;
}

// the deterministic finish of a K-split product: slabs summed in slab order, epilogue applied, A-row-sum partials finished
static int launch_reduce(const ekaid_gemm_t* d, const GemmArgs& a, hipStream_t st) {
    const long long total = (long long)d->M * d->N;
    const bool vec_ok = (d->N % 4 == 0) && (d->ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(d->C) & 15) == 0) &&
                        total / 4 < (1ll << 30);
    if (vec_ok) {
        const int n4 = d->N / 4;
        hipLaunchKernelGGL(splitk_reduce_vec_kernel<4>, dim3(cdiv((long long)d->M * n4, 256)), dim3(256), 0, st, a.ws, d->C,
                           d->M, n4, d->ldc, a.splitk, a.e, a.asum_ws, a.asum, a.splitk * (a.tiles_n < ASUM_SHARE ? a.tiles_n : ASUM_SHARE));
    } else {
        int blocks = cdiv(total, 256);
        if (blocks > 2048) blocks = 2048;
        if (a.asum && (long long)blocks * 256 < d->M) blocks = cdiv(d->M, 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.ws, d->C, d->M, d->N, d->ldc,
                           a.splitk, a.e, a.asum_ws, a.asum, a.splitk * (a.tiles_n < ASUM_SHARE ? a.tiles_n : ASUM_SHARE));
    }
    return check_launch("splitk_reduce_kernel");
}

extern "C" int ekaid_gemm_f32(const ekaid_gemm_t* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !d->A || !d->B || !d->C) { set_error("ekaid_gemm_f32: null pointer"); return EKAID_ERR_ARG; }
    if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch1 < 1 || d->batch2 < 1) {
        set_error("ekaid_gemm_f32: bad dims M=%d N=%d K=%d batch=%dx%d", d->M, d->N, d->K, d->batch1, d->batch2);
        return EKAID_ERR_ARG;
    }
    if (d->transA && d->transB) { set_error("ekaid_gemm_f32: transA && transB unsupported"); return EKAID_ERR_UNSUPPORTED; }
    if (d->rdiv < 1 && d->R) { set_error("ekaid_gemm_f32: rdiv must be >= 1"); return EKAID_ERR_ARG; }
    if (d->post && (d->batch1 * d->batch2 != 1 || d->ldpost < d->N)) {      // `post` is indexed m * ldpost + n: it has no batch strides
        set_error("ekaid_gemm_f32: post needs one batch and ldpost >= N (batch %dx%d, ldpost %d, N %d)", d->batch1, d->batch2, d->ldpost, d->N);
        return EKAID_ERR_ARG;
    }
    if (d->M == 0 || d->N == 0) return EKAID_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        int G, NO;
        if (skinny_tn(d, G, NO)) {
            const size_t need = (size_t)G * d->M * d->N * sizeof(float);
            if (!workspace || workspace_bytes < need) {
                set_error("ekaid_gemm_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
                return EKAID_ERR_ARG;
            }
            const int rows_per_wg = cdiv(d->K, G);
            if (d->M <= 4 && d->N >= 32)
                hipLaunchKernelGGL(tn_skinny_cols_kernel<4>, dim3(cdiv(d->K, rows_per_wg)), dim3(256), 0, st, d->A, d->lda, d->B, d->ldb,
                                   d->M, d->N, d->K, rows_per_wg, NO, (float*)workspace);
            else
                hipLaunchKernelGGL(tn_skinny_kernel, dim3(cdiv(d->K, rows_per_wg)), dim3(256), 0, st, d->A, d->lda, d->B, d->ldb, d->M, d->N,
                                   d->K, rows_per_wg, NO, (float*)workspace);
            hipLaunchKernelGGL(tn_skinny_finish_kernel, dim3(d->M * d->N), dim3(256), 0, st, (const float*)workspace,
                               cdiv(d->K, rows_per_wg), d->M * d->N, d->N, d->C, d->ldc, d->alpha, d->accumulate);
            const int rc_s = check_launch("tn_skinny_kernel");
            if (rc_s || !d->a_rowsum) return rc_s;
            return ekaid_colsum_acc(d->A, d->K, d->M, d->lda, d->a_rowsum, stream);
        }
    }
    GemmArgs a;
    int TMN;
    {
        const int rc0 = fill_args(d, a, TMN, workspace, workspace_bytes, 0);
        if (rc0) return rc0;
    }
    const int batch = d->batch1 * d->batch2;
    const bool akc = !d->transA, bkc = d->transB != 0;
    // a_rowsum (the bias gradient of a weight-gradient product): folded into the A-operand staging of the split-arithmetic
    // kernels -- the column sums read dY a second time otherwise (73 launches, 0.75 ms per step at B = 256) --, by a separate
    // column-sum launch where that kernel does not apply
    bool rowsum_after = false;
    if (d->a_rowsum) {
        if (akc) { set_error("ekaid_gemm_f32: a_rowsum needs transA (A stored [K][M])"); return EKAID_ERR_UNSUPPORTED; }
        if (gemm_mode() >= 1 && TMN >= 1 && TMN <= 4 && batch == 1) {
            a.asum = d->a_rowsum;
            if (a.splitk > 1) {
                const size_t need = (size_t)a.splitk * d->M * ((size_t)d->N + ASUM_SHARE) * sizeof(float);
                if (workspace_bytes < need) { set_error("ekaid_gemm_f32: workspace too small for a_rowsum (%zu < %zu)", workspace_bytes, need); return EKAID_ERR_ARG; }
                a.asum_ws = a.ws + (size_t)a.splitk * d->M * d->N;
            }
        } else {
            rowsum_after = true;
        }
    }
    // (Split-K with the reduction done by the last workgroup of each tile -- arrival counter, device-scope slab accesses --
    // instead of the second kernel was measured and removed: bit-identical, but 0.97-2.9x the time of GEMM + reduce kernel on
    // the recurrent shapes (the reduce kernel uses every CU, the last arriver is one workgroup per tile), and 5x with
    // __threadfence() per workgroup.  profiles/r02_splitk_in_kernel_reduce.txt)
    // (Two ways of repairing the short last wave of the 18432x1024 outputs -- 1152 tiles on 768 resident workgroups -- were
    // measured and removed: the tail rows as 128x64 tiles, 224 -> 232-245 us, and the tail rows split along K into one more
    // full wave, 222 -> 231 us.  With 1.5 workgroups per CU instead of 3 each one runs almost twice as fast: the kernel is
    // bound by what a CU can sustain, not by the tail.  profiles/r02_planes_gemm_microbench.txt)
    int rc = launch_product(a, TMN, akc, bkc, batch, st);
    if (rc) return rc;
    if (a.splitk > 1) rc = launch_reduce(d, a, st);
    if (rc == EKAID_OK && rowsum_after) {
        if (batch != 1) { set_error("ekaid_gemm_f32: a_rowsum with a batched product"); return EKAID_ERR_UNSUPPORTED; }
        rc = ekaid_colsum_acc(d->A, d->K, d->M, d->lda, d->a_rowsum, stream);
    }
    return rc;
}


// ---- raw partial products for consumers that finish the reduction themselves (include/ekaid_hip.h: ekaid_gemm_slabs)
static int slabs_check(const ekaid_gemm_t* d, const char* who) {
    if (!d || !d->A || !d->B) { set_error("%s: null pointer", who); return EKAID_ERR_ARG; }
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) { set_error("%s: bad dims M=%d N=%d K=%d", who, d->M, d->N, d->K); return EKAID_ERR_ARG; }
    if (d->batch1 != 1 || d->batch2 != 1 || (d->transA && d->transB)) { set_error("%s: one batch, not both operands transposed", who); return EKAID_ERR_UNSUPPORTED; }
    return EKAID_OK;
}
extern "C" int ekaid_gemm_slab_count(const ekaid_gemm_t* d) {
    if (slabs_check(d, "ekaid_gemm_slab_count")) return EKAID_ERR_ARG;
    int TMN, splitk, kps, tm, tn;
    t_plan_slab = 1;
    plan_full(d, TMN, splitk, kps, tm, tn);
    t_plan_slab = 0;
    return splitk;
}
extern "C" int ekaid_gemm_slabs(const ekaid_gemm_t* d, float* slabs, size_t slab_bytes, void* stream) {
    int rc = slabs_check(d, "ekaid_gemm_slabs");
    if (rc) return rc;
    GemmArgs a;
    int TMN;
    rc = fill_args(d, a, TMN, slabs, slab_bytes, 1);
    if (rc) return rc;
    return launch_product(a, TMN, !d->transA, d->transB != 0, 1, (hipStream_t)stream);
}
extern "C" int ekaid_gemm_slabs2(const ekaid_gemm_t* d0, float* slabs0, size_t slab_bytes0, const ekaid_gemm_t* d1, float* slabs1,
                                 size_t slab_bytes1, void* stream) {
    int rc = slabs_check(d0, "ekaid_gemm_slabs2");
    if (!rc) rc = slabs_check(d1, "ekaid_gemm_slabs2");
    if (rc) return rc;
    GemmArgs a0, a1;
    int T0, T1;
    rc = fill_args(d0, a0, T0, slabs0, slab_bytes0, 1);
    if (!rc) rc = fill_args(d1, a1, T1, slabs1, slab_bytes1, 1);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool same = (d0->transA != 0) == (d1->transA != 0) && (d0->transB != 0) == (d1->transB != 0);
    if (gemm_mode() < 1 || T0 != T1 || (T0 != 1 && T0 != 2) || !same) {  // outside the pair kernel's envelope: two launches, same results
        rc = launch_product(a0, T0, !d0->transA, d0->transB != 0, 1, st);
        if (rc) return rc;
        return launch_product(a1, T1, !d1->transA, d1->transB != 0, 1, st);
    }
    const bool akc = !d0->transA, bkc = d0->transB != 0;
    count_launch(EKAID_COUNT_GEMM_SPLIT3); count_launch(EKAID_COUNT_GEMM_SPLIT3);     // (slab products carry no operand bounds)
    return pair_dispatch<3>(a0, a1, T0 == 2, akc, bkc, st);
}

// ---- two independent products in one launch (include/ekaid_hip.h: ekaid_gemm_pair_f32)
static size_t al256p(size_t x) { return (x + 255) & ~(size_t)255; }
static bool pair_basic_ok(const ekaid_gemm_t* d) {
    return d && d->A && d->B && d->C && d->M > 0 && d->N > 0 && d->K > 0 && d->batch1 == 1 && d->batch2 == 1 && !(d->transA && d->transB);
}
// can d0 and d1 share a launch?  Same arithmetic family (split kernels), same operand layout, the same square tile from the planner for
// both, no tall-skinny reduction form, A row sums only in the layout whose staging can carry them.
static bool pairable(const ekaid_gemm_t* d0, const ekaid_gemm_t* d1, int& TMN) {
    if (gemm_mode() < 1 || !pair_basic_ok(d0) || !pair_basic_ok(d1)) return false;
    if ((d0->transA != 0) != (d1->transA != 0) || (d0->transB != 0) != (d1->transB != 0)) return false;
    int G, NO;
    if (skinny_tn(d0, G, NO) || skinny_tn(d1, G, NO)) return false;
    if ((d0->a_rowsum || d1->a_rowsum) && !d0->transA) return false;
    // (the row-sum pair kernel exists on the pipelined K loop only: under the tuning override EKAID_GEMM_BIG=plain a solo product of
    //  this layout takes the plain loop, and a pair must stay bit-identical to its two solo launches -- ADVICE r05)
    if ((d0->a_rowsum || d1->a_rowsum) && big_arm(false, false) == 1) return false;
    if ((d0->post && d0->ldpost < d0->N) || (d1->post && d1->ldpost < d1->N) || (d0->R && d0->rdiv < 1) || (d1->R && d1->rdiv < 1)) return false;
    // mode 2: both on the two-term f16 split or both on the exact split (a product inside a pair = the same product alone, bit for bit)
    if (gemm_mode() == 2 && ((d0->a_amax && d0->b_amax) != (d1->a_amax && d1->b_amax))) return false;
    int T0, T1, sk, kps, tm, tn;
    plan_full(d0, T0, sk, kps, tm, tn);
    plan_full(d1, T1, sk, kps, tm, tn);
    if (T0 != T1 || (T0 != 1 && T0 != 2)) return false;
    TMN = T0;
    return true;
}
extern "C" size_t ekaid_gemm_pair_workspace_bytes(const ekaid_gemm_t* d0, const ekaid_gemm_t* d1) {
    const size_t w0 = ekaid_gemm_workspace_bytes(d0), w1 = ekaid_gemm_workspace_bytes(d1);
    int TMN;
    if (pairable(d0, d1, TMN)) return al256p(w0) + w1;         // both slab sets are alive at once
    return w0 > w1 ? w0 : w1;                                   // two launches, one after the other
}
extern "C" int ekaid_gemm_pair_f32(const ekaid_gemm_t* d0, const ekaid_gemm_t* d1, void* workspace, size_t workspace_bytes, void* stream) {
    int TMN = 0;
    if (!pairable(d0, d1, TMN)) {          // outside the pair kernel's envelope: two launches, same results
        const int rc = ekaid_gemm_f32(d0, workspace, workspace_bytes, stream);
        if (rc) return rc;
        return ekaid_gemm_f32(d1, workspace, workspace_bytes, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t w0 = ekaid_gemm_workspace_bytes(d0), w1 = ekaid_gemm_workspace_bytes(d1);
    if ((w0 || w1) && (!workspace || workspace_bytes < al256p(w0) + w1)) {
        set_error("ekaid_gemm_pair_f32: workspace too small (%zu < %zu)", workspace_bytes, al256p(w0) + w1);
        return EKAID_ERR_ARG;
    }
    const ekaid_gemm_t* ds[2] = {d0, d1};
    void* wss[2] = {workspace, (char*)workspace + al256p(w0)};
    const size_t wbs[2] = {w0, w1};
    GemmArgs a[2];
    for (int i = 0; i < 2; ++i) {
        int T;
        const int rc = fill_args(ds[i], a[i], T, wss[i], wbs[i], 0);
        if (rc) return rc;
        if (ds[i]->a_rowsum) {
            a[i].asum = ds[i]->a_rowsum;
            if (a[i].splitk > 1) a[i].asum_ws = a[i].ws + (size_t)a[i].splitk * ds[i]->M * ds[i]->N;
        }
    }
    const bool akc = !d0->transA, bkc = d0->transB != 0;
    const bool x2 = use_x2(a[0]) && use_x2(a[1]);
    int rc = x2 ? pair_dispatch_x2(a[0], a[1], TMN == 2, akc, bkc, st) : pair_dispatch<3>(a[0], a[1], TMN == 2, akc, bkc, st);
    count_launch(x2 ? EKAID_COUNT_GEMM_SPLIT2 : EKAID_COUNT_GEMM_SPLIT3); count_launch(x2 ? EKAID_COUNT_GEMM_SPLIT2 : EKAID_COUNT_GEMM_SPLIT3);
    for (int i = 0; i < 2 && rc == EKAID_OK; ++i)
        if (a[i].splitk > 1) rc = launch_reduce(ds[i], a[i], st);
    return rc;
}
#endif      // EKAID_GEMM_X2_TU
