// Recurrent steps whose consumers finish the split-K reduction themselves.
//
// The decoder (DynamicCore.forward, dynamic_speaker_change_pos.py:94-131) and the question GRU (language_model.py:106-115)
// are chains of small GEMMs (M = batch rows) separated by point-wise / row-local work.  Each of those GEMMs is cut along K
// into slabs (the grid of an M = 32..256 product cannot fill 256 CUs otherwise); summing the slabs used to be a kernel of
// its own between the GEMM and its consumer -- 255 launches per training step.  Here every consumer reads the slabs
// (ekaid_gemm_slabs) directly, in a fixed order (deterministic), adds the bias and goes on with its own arithmetic, and the
// row-local chains of one decoder step are merged into ONE kernel per direction:
//   spk_step_a_fwd : att-LSTM cell -> module weights (fc, softmax over 3) -> attended feature -> POS chain
//   spk_step_a_bwd : the same chain backwards (module mix, att-LSTM cell, POS chain)
// so that a decoder step is 4 GEMM launches + 4 consumers forward and backward (was 15 + 4 reduces / 16 + 5 reduces).
#include "common.h"
#include "../../include/ekaid_hip.h"

namespace ekaid {

#define ST ((hipStream_t)stream)
#define INVK(p) ((p) > 0.f ? 1.f / (1.f - (p)) : 1.f)
constexpr int STEP_MAXR = 2048;     // staged row length limit (rnn_size) of the row-local kernels
constexpr int STEP_MAXP = 64;       // POS classes limit

struct Slabs { const float* p; int n; long long stride; int ld; };
struct Slabs3 { Slabs s[3]; int n; };

static inline Slabs to_slabs(const ekaid_slabs_t* s) {
    Slabs r;
    if (!s || !s->p || s->n <= 0) { r.p = nullptr; r.n = 0; r.stride = 0; r.ld = 0; }
    else { r.p = s->p; r.n = s->n; r.stride = s->stride; r.ld = s->ld; }
    return r;
}
static inline bool to_slabs3(const ekaid_slabs_t* src, int nsrc, Slabs3& out) {
    if (nsrc < 0 || nsrc > 3 || (nsrc > 0 && !src)) return false;
    out.n = 0;
    for (int i = 0; i < 3; ++i) { out.s[i].p = nullptr; out.s[i].n = 0; out.s[i].stride = 0; out.s[i].ld = 0; }
    for (int i = 0; i < nsrc; ++i) {
        const Slabs s = to_slabs(src + i);
        if (s.n > 0) out.s[out.n++] = s;
    }
    return true;
}
static inline bool vec_ok(const Slabs3& S, int cols) {
    if (cols & 3) return false;
    for (int i = 0; i < S.n; ++i)
        if ((S.s[i].ld & 3) || (S.s[i].stride & 3) || (reinterpret_cast<uintptr_t>(S.s[i].p) & 15)) return false;
    return true;
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// sum over the slabs of one source at element offset `off` (row * ld + col); four loads in flight
__device__ __forceinline__ float slab_sum(const Slabs& s, long long off) {
    float acc = 0.f;
    const float* q = s.p + off;
    int i = 0;
    for (; i + 4 <= s.n; i += 4) {
        const float a = q[0], b = q[s.stride], c = q[2 * s.stride], d = q[3 * s.stride];
        acc += (a + b) + (c + d);
        q += 4 * s.stride;
    }
    for (; i < s.n; ++i) { acc += q[0]; q += s.stride; }
    return acc;
}
__device__ __forceinline__ f32x4 slab_sum4(const Slabs& s, long long off) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* q = reinterpret_cast<const f32x4*>(s.p + off);
    const long long st = s.stride >> 2;
    int i = 0;
    for (; i + 4 <= s.n; i += 4) {
        const f32x4 a = q[0], b = q[st], c = q[2 * st], d = q[3 * st];
        acc += (a + b) + (c + d);
        q += 4 * st;
    }
    for (; i < s.n; ++i) { acc += q[0]; q += st; }
    return acc;
}
__device__ __forceinline__ float slab_sum_all(const Slabs3& S, long long row, int col) {
    float v = 0.f;
    for (int k = 0; k < S.n; ++k) v += slab_sum(S.s[k], row * S.s[k].ld + col);
    return v;
}
__device__ __forceinline__ f32x4 slab_sum_all4(const Slabs3& S, long long row, int col) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < S.n; ++k) v += slab_sum4(S.s[k], row * S.s[k].ld + col);
    return v;
}

// block-wide sum of NV values per thread (256 threads); result valid in every thread
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [NV][4] */) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 4 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (red[i * 4] + red[i * 4 + 1]) + (red[i * 4 + 2] + red[i * 4 + 3]);
}

// ------------------------------------------------------------------------------------------------------------------
// out[r][c] = dropout(act(sum of all slabs + bias[c]))        (mask index r * cols + c: the GEMM epilogue's convention)
template <bool VEC>
__global__ __launch_bounds__(256) void slab_epilogue_kernel(Slabs3 S, const float* __restrict__ bias, int rows, int cols, int act, float p,
                                                            float inv, DropSeed seed, float* __restrict__ out, int ldo) {
    if (VEC) {
        const int c4n = cols >> 2;
        const long long n = (long long)rows * c4n;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long r = i / c4n;
            const int c = (int)(i - r * c4n) * 4;
            f32x4 v = slab_sum_all4(S, r, c);
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_apply(v[e], act) * dropout_scale(seed, (uint64_t)(r * cols + c + e), p, inv);
            *reinterpret_cast<f32x4*>(out + r * ldo + c) = o;
        }
    } else {
        const long long n = (long long)rows * cols;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long r = i / cols;
            const int c = (int)(i - r * cols);
            const float v = slab_sum_all(S, r, c) + (bias ? bias[c] : 0.f);
            out[r * ldo + c] = act_apply(v, act) * dropout_scale(seed, (uint64_t)i, p, inv);
        }
    }
}

// dy = sum of the slabs; out[r][c] = dy * keep(seed, r*cols + c)/(1-p) * act'(y[r][c])
__global__ __launch_bounds__(256) void drop_act_slabs_bwd_kernel(Slabs S, const float* __restrict__ y, int ldy, int rows, int cols, int act,
                                                                 float p, float inv, DropSeed seed, float* __restrict__ out, int ldo) {
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        const float dy = slab_sum(S, r * S.ld + c);
        out[r * ldo + c] = dy * dropout_scale(seed, (uint64_t)i, p, inv) * act_grad_from_out(y[r * ldy + c], act);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// LSTM cell: gates (i,f,g,o) = gx + sum of all slabs + bias
__device__ __forceinline__ void lstm_point(float pi, float pf, float pg, float po, float cprev, float& ig, float& fg, float& gg, float& og,
                                           float& cn, float& hn) {
    ig = sigm(pi); fg = sigm(pf); gg = tanhf(pg); og = sigm(po);
    cn = fg * cprev + ig * gg;
    hn = og * tanhf(cn);
}
__global__ __launch_bounds__(256) void lstm_cell_slabs_fwd_kernel(const float* __restrict__ gx, Slabs3 S, const float* __restrict__ bias,
                                                                  const float* __restrict__ c, int B, int H, float* __restrict__ h2,
                                                                  float* __restrict__ c2, float* __restrict__ save) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / H;
        const int j = (int)(i - b * H);
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pre[g] = gx[b * 4 * H + g * H + j] + slab_sum_all(S, b, g * H + j) + (bias ? bias[g * H + j] : 0.f);
        float ig, fg, gg, og, cn, hn;
        lstm_point(pre[0], pre[1], pre[2], pre[3], c[i], ig, fg, gg, og, cn, hn);
        c2[i] = cn;
        h2[i] = hn;
        float* sp = save + b * 4 * H;
        sp[j] = ig; sp[H + j] = fg; sp[2 * H + j] = gg; sp[3 * H + j] = og;
    }
}
// upstream gradient on h2 = sum of the carry slabs (may be absent) + dh_b (may be NULL)
__global__ __launch_bounds__(256) void lstm_cell_slabs_bwd_kernel(Slabs carry, const float* __restrict__ dh_b, const float* __restrict__ dc2,
                                                                  const float* __restrict__ save, const float* __restrict__ c,
                                                                  const float* __restrict__ c2, int B, int H, float* __restrict__ dgates,
                                                                  int lddg, float* __restrict__ dc) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / H;
        const int j = (int)(i - b * H);
        const float* sp = save + b * 4 * H;
        const float ig = sp[j], fg = sp[H + j], gg = sp[2 * H + j], og = sp[3 * H + j];
        const float tc = tanhf(c2[i]);
        const float dh = (carry.n ? slab_sum(carry, b * carry.ld + j) : 0.f) + (dh_b ? dh_b[i] : 0.f);
        const float dct = (dc2 ? dc2[i] : 0.f) + dh * og * (1.f - tc * tc);
        float* o = dgates + b * lddg;
        o[j] = dct * gg * ig * (1.f - ig);
        o[H + j] = dct * c[i] * fg * (1.f - fg);
        o[2 * H + j] = dct * ig * (1.f - gg * gg);
        o[3 * H + j] = dh * tc * og * (1.f - og);
        dc[i] = dct * fg;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GRU cell (gate order r,z,n; b_hn inside r*(.)): gh = sum of the slabs + b_hh
__global__ __launch_bounds__(256) void gru_cell_slabs_fwd_kernel(const float* __restrict__ gi, int ldgi, Slabs S, const float* __restrict__ bhh,
                                                                 const float* __restrict__ h, int ldh, int B, int H, float* __restrict__ hnew,
                                                                 int ldhn, float* __restrict__ save) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / H;
        const int j = (int)(i - b * H);
        const float* gip = gi + b * ldgi;
        const float ghr = slab_sum(S, b * S.ld + j) + bhh[j];
        const float ghz = slab_sum(S, b * S.ld + H + j) + bhh[H + j];
        const float ghn = slab_sum(S, b * S.ld + 2 * H + j) + bhh[2 * H + j];
        const float r = sigm(gip[j] + ghr);
        const float z = sigm(gip[H + j] + ghz);
        const float nn = tanhf(gip[2 * H + j] + r * ghn);
        hnew[b * ldhn + j] = (1.f - z) * nn + z * h[b * ldh + j];
        float* sp = save + b * 4 * H;
        sp[j] = r; sp[H + j] = z; sp[2 * H + j] = nn; sp[3 * H + j] = ghn;
    }
}
// upstream = dhnew (strided, may be NULL) + dh_direct (the z * d term of the following step, may be NULL) + carry slabs
// (dgh_{t+1} . W_hh, may be absent).  Writes dgi, dgh and this step's direct term z * d.
__global__ __launch_bounds__(256) void gru_cell_slabs_bwd_kernel(const float* __restrict__ dhnew, int lddh, const float* __restrict__ dh_direct,
                                                                 Slabs carry, const float* __restrict__ save, const float* __restrict__ h,
                                                                 int ldh, int B, int H, float* __restrict__ dgi, int lddgi,
                                                                 float* __restrict__ dgh, float* __restrict__ dh_out) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / H;
        const int j = (int)(i - b * H);
        const float* sp = save + b * 4 * H;
        const float r = sp[j], z = sp[H + j], nn = sp[2 * H + j], ghn = sp[3 * H + j];
        const float d = (dhnew ? dhnew[b * lddh + j] : 0.f) + (dh_direct ? dh_direct[i] : 0.f) +
                        (carry.n ? slab_sum(carry, b * carry.ld + j) : 0.f);
        const float dn = d * (1.f - z);
        const float dz = d * (h[b * ldh + j] - nn);
        const float dpn = dn * (1.f - nn * nn);
        const float dpr = dpn * ghn * r * (1.f - r);
        const float dpz = dz * z * (1.f - z);
        float* gi_o = dgi + b * lddgi;
        float* gh_o = dgh + b * 3 * H;
        gi_o[j] = dpr; gi_o[H + j] = dpz; gi_o[2 * H + j] = dpn;
        gh_o[j] = dpr; gh_o[H + j] = dpz; gh_o[2 * H + j] = dpn * r;
        dh_out[i] = d * z;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gate = sigmoid(sum of the slabs + bias); gated = gate * att     (dynamic_speaker_change_pos.py:122-125)
__global__ __launch_bounds__(256) void gate_mul_slabs_fwd_kernel(Slabs S, const float* __restrict__ bias, const float* __restrict__ att, int B,
                                                                 int A, float* __restrict__ gate, float* __restrict__ gated) {
    const long long n = (long long)B * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / A;
        const int k = (int)(i - b * A);
        const float g = sigm(slab_sum(S, b * S.ld + k) + bias[k]);
        gate[i] = g;
        gated[i] = g * att[i];
    }
}
// d_gated = sum of the slabs (kept for the module-mix backward); dz = d_gated * att * gate * (1 - gate)
__global__ __launch_bounds__(256) void gate_slabs_bwd_kernel(Slabs S, const float* __restrict__ att, const float* __restrict__ gate, int B, int A,
                                                             float* __restrict__ dz, float* __restrict__ d_gated) {
    const long long n = (long long)B * A;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long b = i / A;
        const int k = (int)(i - b * A);
        const float dg = slab_sum(S, b * S.ld + k);
        const float g = gate[i];
        d_gated[i] = dg;
        dz[i] = dg * att[i] * g * (1.f - g);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Row-local first half of a decoder step, one 256-thread workgroup per batch row:
//   1. module-attention LSTM cell: gates = gxa + sum(Satt slabs) + b_hh  -> h_att (also staged in LDS), c_att, gate activations
//   2. module weights mw = softmax(h_att . w_fc^T + b_fc); att = mw0*bef + mw1*diff + mw2*aft -> att (contiguous) and att2
//   3. POS chain: v = dropout(relu(sum(Spos slabs) + b1)); dpos = dropout(w_wp v + b_wp); pw = softmax(dpos); pp = w_p2 pw + b_p2
struct StepAFwd {
    const float* gxa; Slabs3 Satt; const float* bhh; const float* c_prev; int R;
    float* h_out; float* c_out; float* save;
    const float* w_fc; const float* b_fc; const float* bef; const float* diff; const float* aft; int A;
    float* mw; float* att; float* att2; int ld2;
    Slabs Spos; const float* b1; const float* w_wp; const float* b_wp; int P; const float* w_p2; const float* b_p2;
    float pv, invv; DropSeed seed_v; float p5, inv5; DropSeed seed_d;
    float* vp; float* dp; float* pw; float* pp; int ldpp;
};
__global__ __launch_bounds__(256) void spk_step_a_fwd_kernel(StepAFwd a) {
    // Two workgroups per batch row: blockIdx.y = 0 runs the cell + module-weight chain, 1 the POS chain (it depends on the
    // h_lang product only, not on the cell), so the two dependent chains of a row overlap instead of adding up.
    __shared__ float sh[STEP_MAXR];     // h_att row (y = 0) / the POS vector v (y = 1)
    __shared__ float sz[STEP_MAXP];
    __shared__ float red[12];
    const int b = blockIdx.x, tid = threadIdx.x, R = a.R, A = a.A, P = a.P;
    if (blockIdx.y == 0) {
        // 1. LSTM cell
        for (int j = tid; j < R; j += 256) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = a.gxa[(long long)b * 4 * R + g * R + j] + slab_sum_all(a.Satt, b, g * R + j) + a.bhh[g * R + j];
            float ig, fg, gg, og, cn, hn;
            const long long i = (long long)b * R + j;
            lstm_point(pre[0], pre[1], pre[2], pre[3], a.c_prev[i], ig, fg, gg, og, cn, hn);
            a.c_out[i] = cn;
            a.h_out[i] = hn;
            sh[j] = hn;
            float* sp = a.save + (long long)b * 4 * R;
            sp[j] = ig; sp[R + j] = fg; sp[2 * R + j] = gg; sp[3 * R + j] = og;
        }
        __syncthreads();
        // 2. module weights + attended feature
        float z[3] = {0.f, 0.f, 0.f};
        for (int k = tid; k < R; k += 256) {
            const float x = sh[k];
            z[0] += x * a.w_fc[k]; z[1] += x * a.w_fc[R + k]; z[2] += x * a.w_fc[2 * R + k];
        }
        block_sum<3>(z, red);
        z[0] += a.b_fc[0]; z[1] += a.b_fc[1]; z[2] += a.b_fc[2];
        const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
        const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
        const float inv = 1.f / (e0 + e1 + e2);
        const float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
        if (tid == 0) { a.mw[b * 3] = w0; a.mw[b * 3 + 1] = w1; a.mw[b * 3 + 2] = w2; }
        for (int k = tid; k < A; k += 256) {
            const long long i = (long long)b * A + k;
            const float v = w0 * a.bef[i] + w1 * a.diff[i] + w2 * a.aft[i];
            a.att[i] = v;
            if (a.att2) a.att2[(long long)b * a.ld2 + k] = v;
        }
        return;
    }
    // 3. POS chain
    for (int k = tid; k < R; k += 256) {
        float v = slab_sum(a.Spos, (long long)b * a.Spos.ld + k) + a.b1[k];
        v = (v > 0.f ? v : 0.f) * dropout_scale(a.seed_v, (uint64_t)b * R + k, a.pv, a.invv);
        sh[k] = v;
        a.vp[(long long)b * R + k] = v;
    }
    __syncthreads();
    const int l = tid & 15, jg = tid >> 4;
    for (int j0 = 0; j0 < P; j0 += 16) {
        const int j = j0 + jg;
        float s = 0.f;
        if (j < P) {
            const float* wr = a.w_wp + (long long)j * R;
            for (int k = l; k < R; k += 16) s += wr[k] * sh[k];
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
        if (j < P && l == 0) {
            const float zz = (s + a.b_wp[j]) * dropout_scale(a.seed_d, (uint64_t)b * P + j, a.p5, a.inv5);
            sz[j] = zz;
            a.dp[(long long)b * P + j] = zz;
        }
    }
    __syncthreads();
    if (tid < 64) {                      // softmax over the P <= 64 classes: one lane per class
        const float x = tid < P ? sz[tid] : -INFINITY;
        const float m2 = wave_max(x);
        const float e = tid < P ? __expf(x - m2) : 0.f;
        const float sum = wave_sum(e);
        if (tid < P) {
            const float w = e / sum;
            sz[tid] = w;
            a.pw[(long long)b * P + tid] = w;
        }
    }
    __syncthreads();
    for (int k = tid; k < R; k += 256) {
        const float* wr = a.w_p2 + (long long)k * P;
        float s = a.b_p2[k];
        for (int j = 0; j < P; ++j) s += wr[j] * sz[j];
        a.pp[(long long)b * a.ldpp + k] = s;
    }
}

// The same chain backwards, one workgroup per batch row:
//   1. module mix: d_att = Sdpa[:, R + k] + d_gated * gate;  d{bef,diff,aft} += mw_c * d_att;  dz = softmax'(mw) <d_att, feat_c>
//   2. att-LSTM cell: dh = sum(Sdha slabs, the carry) + dz . w_fc;  -> dgates (leading dim lddg), dc
//   3. POS chain from dpp = Sdpa[:, k < R] (also written to dpp_out for the pos2 weight gradient)
struct StepABwd {
    Slabs Sdpa; const float* d_gated; const float* gate; const float* mw; const float* bef; const float* diff; const float* aft; int A;
    const float* w_fc; int R; float* dz_fc; float* d_bef; float* d_diff; float* d_aft;
    Slabs Sdha; const float* dc_in; const float* save; const float* c_prev; const float* c_cur; float* dgates; int lddg; float* dc_out;
    const float* pw; const float* dhead; int P; const float* w_p2; const float* w_wp; const float* vp;
    float pv, invv; DropSeed seed_v; float p5, inv5; DropSeed seed_d;
    float* dpp_out; int lddpp; float* dwp_out; float* dx_out; int lddx;
};
__global__ __launch_bounds__(256) void spk_step_a_bwd_kernel(StepABwd a) {
    // two workgroups per batch row, as in the forward kernel: blockIdx.y = 0 module mix + cell, 1 the POS chain
    __shared__ float sd[STEP_MAXR];
    __shared__ float sg[STEP_MAXP];
    __shared__ float sp[256];
    __shared__ float red[12];
    const int b = blockIdx.x, tid = threadIdx.x, R = a.R, A = a.A, P = a.P;
    if (blockIdx.y == 0) {
        // 1. module mix
        const float w0 = a.mw[b * 3], w1 = a.mw[b * 3 + 1], w2 = a.mw[b * 3 + 2];
        float s[3] = {0.f, 0.f, 0.f};
        for (int k = tid; k < A; k += 256) {
            const long long i = (long long)b * A + k;
            const float d = slab_sum(a.Sdpa, (long long)b * a.Sdpa.ld + R + k) + a.d_gated[i] * a.gate[i];
            s[0] += d * a.bef[i]; s[1] += d * a.diff[i]; s[2] += d * a.aft[i];
            a.d_bef[i] += w0 * d; a.d_diff[i] += w1 * d; a.d_aft[i] += w2 * d;
        }
        block_sum<3>(s, red);
        const float dot = w0 * s[0] + w1 * s[1] + w2 * s[2];
        const float dz0 = w0 * (s[0] - dot), dz1 = w1 * (s[1] - dot), dz2 = w2 * (s[2] - dot);
        if (tid == 0) { a.dz_fc[b * 3] = dz0; a.dz_fc[b * 3 + 1] = dz1; a.dz_fc[b * 3 + 2] = dz2; }
        // 2. att-LSTM cell
        for (int j = tid; j < R; j += 256) {
            const long long i = (long long)b * R + j;
            const float dh = (a.Sdha.n ? slab_sum(a.Sdha, (long long)b * a.Sdha.ld + j) : 0.f) +
                             dz0 * a.w_fc[j] + dz1 * a.w_fc[R + j] + dz2 * a.w_fc[2 * R + j];
            const float* sv = a.save + (long long)b * 4 * R;
            const float ig = sv[j], fg = sv[R + j], gg = sv[2 * R + j], og = sv[3 * R + j];
            const float tc = tanhf(a.c_cur[i]);
            const float dct = a.dc_in[i] + dh * og * (1.f - tc * tc);
            float* o = a.dgates + (long long)b * a.lddg;
            o[j] = dct * gg * ig * (1.f - ig);
            o[R + j] = dct * a.c_prev[i] * fg * (1.f - fg);
            o[2 * R + j] = dct * ig * (1.f - gg * gg);
            o[3 * R + j] = dh * tc * og * (1.f - og);
            a.dc_out[i] = dct * fg;
        }
        return;
    }
    // 3. POS chain
    for (int k = tid; k < R; k += 256) {
        const float v = slab_sum(a.Sdpa, (long long)b * a.Sdpa.ld + k);
        sd[k] = v;
        a.dpp_out[(long long)b * a.lddpp + k] = v;
    }
    __syncthreads();
    {
        const int PG = P <= 16 ? 16 : (P <= 32 ? 32 : 64);
        const int j = tid % PG, kq = tid / PG, KQ = 256 / PG;
        float t = 0.f;
        if (j < P)
            for (int k = kq; k < R; k += KQ) t += sd[k] * a.w_p2[(long long)k * P + j];
        sp[tid] = t;
        __syncthreads();
        if (tid < P) {
            float u = 0.f;
            for (int q = 0; q < KQ; ++q) u += sp[q * PG + tid];
            sg[tid] = u;
        }
    }
    __syncthreads();
    if (tid < 64) {                      // softmax backward over the P <= 64 classes: one lane per class
        const float pwj = tid < P ? a.pw[(long long)b * P + tid] : 0.f;
        const float sgj = tid < P ? sg[tid] : 0.f;
        const float dt = wave_sum(pwj * sgj);
        if (tid < P) {
            float g = pwj * (sgj - dt);
            if (a.dhead) g += a.dhead[(long long)b * P + tid];
            g *= dropout_scale(a.seed_d, (uint64_t)b * P + tid, a.p5, a.inv5);
            sg[tid] = g;
            a.dwp_out[(long long)b * P + tid] = g;
        }
    }
    __syncthreads();
    for (int k = tid; k < R; k += 256) {
        float t = 0.f;
        for (int j = 0; j < P; ++j) t += sg[j] * a.w_wp[(long long)j * R + k];
        const float y = a.vp[(long long)b * R + k];
        a.dx_out[(long long)b * a.lddx + k] = t * dropout_scale(a.seed_v, (uint64_t)b * R + k, a.pv, a.invv) * (y > 0.f ? 1.f : 0.f);
    }
}

static inline dim3 g1(long long n) {
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

}  // namespace ekaid

using namespace ekaid;

extern "C" int ekaid_slab_epilogue(const ekaid_slabs_t* src, int nsrc, const float* bias, int rows, int cols, int act, float p,
                                   uint64_t seed, float* out, int ldo, void* stream) {
    Slabs3 S;
    if (!to_slabs3(src, nsrc, S) || !out) { set_error("slab_epilogue: up to 3 sources, non-null output"); return EKAID_ERR_ARG; }
    if (p < 0.f || p >= 1.f) { set_error("slab_epilogue: p=%f out of range", p); return EKAID_ERR_ARG; }
    if (rows <= 0 || cols <= 0) return EKAID_OK;
    const bool vec = vec_ok(S, cols) && (ldo % 4 == 0) && al16(out) && (!bias || al16(bias));
    if (vec)
        hipLaunchKernelGGL((slab_epilogue_kernel<true>), g1((long long)rows * (cols / 4)), dim3(256), 0, ST, S, bias, rows, cols, act, p, INVK(p),
                           DS(seed), out, ldo);
    else
        hipLaunchKernelGGL((slab_epilogue_kernel<false>), g1((long long)rows * cols), dim3(256), 0, ST, S, bias, rows, cols, act, p, INVK(p),
                           DS(seed), out, ldo);
    return check_launch("slab_epilogue");
}

extern "C" int ekaid_drop_act_slabs_bwd(const ekaid_slabs_t* src, const float* y, int ldy, int rows, int cols, int act, float p,
                                        uint64_t seed, float* out, int ldo, void* stream) {
    const Slabs S = to_slabs(src);
    if (!S.n || !y || !out) { set_error("drop_act_slabs_bwd: null argument"); return EKAID_ERR_ARG; }
    if (rows <= 0 || cols <= 0) return EKAID_OK;
    hipLaunchKernelGGL(drop_act_slabs_bwd_kernel, g1((long long)rows * cols), dim3(256), 0, ST, S, y, ldy, rows, cols, act, p, INVK(p), DS(seed),
                       out, ldo);
    return check_launch("drop_act_slabs_bwd");
}

extern "C" int ekaid_lstm_cell_slabs_fwd(const float* gx, const ekaid_slabs_t* src, int nsrc, const float* bias, const float* c, int B,
                                         int H, float* h2, float* c2, float* save, void* stream) {
    Slabs3 S;
    if (!to_slabs3(src, nsrc, S) || !gx || !c || !h2 || !c2 || !save) { set_error("lstm_cell_slabs_fwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || H <= 0) return EKAID_OK;
    hipLaunchKernelGGL(lstm_cell_slabs_fwd_kernel, g1((long long)B * H), dim3(256), 0, ST, gx, S, bias, c, B, H, h2, c2, save);
    return check_launch("lstm_cell_slabs_fwd");
}

extern "C" int ekaid_lstm_cell_slabs_bwd(const ekaid_slabs_t* carry, const float* dh_b, const float* dc2, const float* save,
                                         const float* c, const float* c2, int B, int H, float* dgates, int lddg, float* dc, void* stream) {
    if (!save || !c || !c2 || !dgates || !dc) { set_error("lstm_cell_slabs_bwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || H <= 0) return EKAID_OK;
    hipLaunchKernelGGL(lstm_cell_slabs_bwd_kernel, g1((long long)B * H), dim3(256), 0, ST, to_slabs(carry), dh_b, dc2, save, c, c2, B, H, dgates,
                       lddg, dc);
    return check_launch("lstm_cell_slabs_bwd");
}

extern "C" int ekaid_gru_cell_slabs_fwd(const float* gi, int ldgi, const ekaid_slabs_t* gh, const float* b_hh, const float* h, int ldh,
                                        int B, int H, float* hnew, int ldhn, float* save, void* stream) {
    const Slabs S = to_slabs(gh);
    if (!S.n || !gi || !b_hh || !h || !hnew || !save) { set_error("gru_cell_slabs_fwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || H <= 0) return EKAID_OK;
    hipLaunchKernelGGL(gru_cell_slabs_fwd_kernel, g1((long long)B * H), dim3(256), 0, ST, gi, ldgi, S, b_hh, h, ldh, B, H, hnew, ldhn, save);
    return check_launch("gru_cell_slabs_fwd");
}

extern "C" int ekaid_gru_cell_slabs_bwd(const float* dhnew, int lddh, const float* dh_direct, const ekaid_slabs_t* carry, const float* save,
                                        const float* h, int ldh, int B, int H, float* dgi, int lddgi, float* dgh, float* dh_out,
                                        void* stream) {
    if (!save || !h || !dgi || !dgh || !dh_out) { set_error("gru_cell_slabs_bwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || H <= 0) return EKAID_OK;
    hipLaunchKernelGGL(gru_cell_slabs_bwd_kernel, g1((long long)B * H), dim3(256), 0, ST, dhnew, lddh, dh_direct, to_slabs(carry), save, h, ldh,
                       B, H, dgi, lddgi, dgh, dh_out);
    return check_launch("gru_cell_slabs_bwd");
}

extern "C" int ekaid_gate_mul_slabs_fwd(const ekaid_slabs_t* src, const float* bias, const float* att, int B, int A, float* gate,
                                        float* gated, void* stream) {
    const Slabs S = to_slabs(src);
    if (!S.n || !bias || !att || !gate || !gated) { set_error("gate_mul_slabs_fwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || A <= 0) return EKAID_OK;
    hipLaunchKernelGGL(gate_mul_slabs_fwd_kernel, g1((long long)B * A), dim3(256), 0, ST, S, bias, att, B, A, gate, gated);
    return check_launch("gate_mul_slabs_fwd");
}

extern "C" int ekaid_gate_slabs_bwd(const ekaid_slabs_t* src, const float* att, const float* gate, int B, int A, float* dz,
                                    float* d_gated, void* stream) {
    const Slabs S = to_slabs(src);
    if (!S.n || !att || !gate || !dz || !d_gated) { set_error("gate_slabs_bwd: null argument"); return EKAID_ERR_ARG; }
    if (B <= 0 || A <= 0) return EKAID_OK;
    hipLaunchKernelGGL(gate_slabs_bwd_kernel, g1((long long)B * A), dim3(256), 0, ST, S, att, gate, B, A, dz, d_gated);
    return check_launch("gate_slabs_bwd");
}

extern "C" int ekaid_spk_step_a_fwd(const ekaid_spk_step_a_fwd_t* q, void* stream) {
    if (!q) { set_error("spk_step_a_fwd: null descriptor"); return EKAID_ERR_ARG; }
    if (q->R > STEP_MAXR || q->P > STEP_MAXP || q->R < 1 || q->P < 1) {
        set_error("ekaid_spk_step_a_fwd: 1 <= R <= %d and 1 <= P <= %d required", STEP_MAXR, STEP_MAXP);
        return EKAID_ERR_UNSUPPORTED;
    }
    if (q->B <= 0) return EKAID_OK;
    StepAFwd a;
    a.gxa = q->gxa;
    if (!to_slabs3(q->gates_src, q->n_gates_src, a.Satt)) { set_error("spk_step_a_fwd: up to 3 gate sources"); return EKAID_ERR_ARG; }
    a.bhh = q->b_hh; a.c_prev = q->c_prev; a.R = q->R; a.h_out = q->h_out; a.c_out = q->c_out; a.save = q->gate_save;
    a.w_fc = q->w_fc; a.b_fc = q->b_fc; a.bef = q->bef; a.diff = q->diff; a.aft = q->aft; a.A = q->A;
    a.mw = q->mw; a.att = q->att; a.att2 = q->att2; a.ld2 = q->ld2;
    a.Spos = to_slabs(&q->pos_src); a.b1 = q->b_p1; a.w_wp = q->w_wp; a.b_wp = q->b_wp; a.P = q->P; a.w_p2 = q->w_p2; a.b_p2 = q->b_p2;
    a.pv = q->p_v; a.invv = INVK(q->p_v); a.seed_v = DS(q->seed_v); a.p5 = q->p_d; a.inv5 = INVK(q->p_d); a.seed_d = DS(q->seed_d);
    a.vp = q->vp; a.dp = q->dp; a.pw = q->pw; a.pp = q->pp; a.ldpp = q->ldpp;
    if (!a.gxa || !a.bhh || !a.c_prev || !a.h_out || !a.c_out || !a.save || !a.w_fc || !a.b_fc || !a.bef || !a.diff || !a.aft || !a.mw ||
        !a.att || !a.Spos.n || !a.b1 || !a.w_wp || !a.b_wp || !a.w_p2 || !a.b_p2 || !a.vp || !a.dp || !a.pw || !a.pp) {
        set_error("spk_step_a_fwd: null pointer in the descriptor");
        return EKAID_ERR_ARG;
    }
    hipLaunchKernelGGL(spk_step_a_fwd_kernel, dim3(q->B, 2), dim3(256), 0, ST, a);
    return check_launch("spk_step_a_fwd");
}

extern "C" int ekaid_spk_step_a_bwd(const ekaid_spk_step_a_bwd_t* q, void* stream) {
    if (!q) { set_error("spk_step_a_bwd: null descriptor"); return EKAID_ERR_ARG; }
    if (q->R > STEP_MAXR || q->P > STEP_MAXP || q->R < 1 || q->P < 1) {
        set_error("ekaid_spk_step_a_bwd: 1 <= R <= %d and 1 <= P <= %d required", STEP_MAXR, STEP_MAXP);
        return EKAID_ERR_UNSUPPORTED;
    }
    if (q->B <= 0) return EKAID_OK;
    StepABwd a;
    a.Sdpa = to_slabs(&q->dpa_src); a.d_gated = q->d_gated; a.gate = q->gate; a.mw = q->mw; a.bef = q->bef; a.diff = q->diff; a.aft = q->aft;
    a.A = q->A; a.w_fc = q->w_fc; a.R = q->R; a.dz_fc = q->dz_fc; a.d_bef = q->d_bef; a.d_diff = q->d_diff; a.d_aft = q->d_aft;
    a.Sdha = to_slabs(&q->dh_carry); a.dc_in = q->dc_in; a.save = q->gate_save; a.c_prev = q->c_prev; a.c_cur = q->c_cur;
    a.dgates = q->dgates; a.lddg = q->lddg; a.dc_out = q->dc_out;
    a.pw = q->pw; a.dhead = q->dhead; a.P = q->P; a.w_p2 = q->w_p2; a.w_wp = q->w_wp; a.vp = q->vp;
    a.pv = q->p_v; a.invv = INVK(q->p_v); a.seed_v = DS(q->seed_v); a.p5 = q->p_d; a.inv5 = INVK(q->p_d); a.seed_d = DS(q->seed_d);
    a.dpp_out = q->dpp_out; a.lddpp = q->lddpp; a.dwp_out = q->dwp_out; a.dx_out = q->dx_out; a.lddx = q->lddx;
    if (!a.Sdpa.n || !a.d_gated || !a.gate || !a.mw || !a.bef || !a.diff || !a.aft || !a.w_fc || !a.dz_fc || !a.d_bef || !a.d_diff ||
        !a.d_aft || !a.dc_in || !a.save || !a.c_prev || !a.c_cur || !a.dgates || !a.dc_out || !a.pw || !a.w_p2 || !a.w_wp || !a.vp ||
        !a.dpp_out || !a.dwp_out || !a.dx_out) {
        set_error("spk_step_a_bwd: null pointer in the descriptor");
        return EKAID_ERR_ARG;
    }
    hipLaunchKernelGGL(spk_step_a_bwd_kernel, dim3(q->B, 2), dim3(256), 0, ST, a);
    return check_launch("spk_step_a_bwd");
}
