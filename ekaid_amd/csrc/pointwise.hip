// HBM-bound point-wise / reduction kernels of the path.  Grid-stride, 256-thread workgroups, at most
// 2048 workgroups (8 per CU); 16-byte accesses where the tensor allows it.
#include "common.h"
#include "../../include/ekaid_hip.h"

namespace ekaid {

static inline dim3 grid1d(long long n, int per_thread = 1) {
    long long b = (n + 256ll * per_thread - 1) / (256ll * per_thread);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}
#define GS_LOOP(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ weight norm
__global__ __launch_bounds__(256) void sumsq_kernel(const float* v, long long n, double* acc) {
    double s = 0.0;
    GS_LOOP(i, n) { double x = v[i]; s += x * x; }
    s = wave_sum_f64(s);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) acc[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);      // one partial per workgroup: no atomics
}
// sum of `n` per-workgroup partials in a fixed order (deterministic); result valid in thread 0
__device__ __forceinline__ double sum_partials_256(const double* part, int n) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum_f64(s);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    return (w[0] + w[1]) + (w[2] + w[3]);
}
__global__ __launch_bounds__(256) void wn_finish_kernel(const double* part, int nparts, const float* g, float* scale_out, float* norm_out) {
    const double tot = sum_partials_256(part, nparts);
    if (threadIdx.x == 0) {
        const float nrm = (float)sqrt(tot);
        norm_out[0] = nrm;
        scale_out[0] = g[0] / nrm;
    }
}
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* v, const float* g, const float* norm,
                                                     const float* dscale, long long n, float* dv, int accumulate,
                                                     float* dg) {
    const float nrm = norm[0];
    const float coef = -dscale[0] * g[0] / (nrm * nrm * nrm);
    GS_LOOP(i, n) {
        const float t = coef * v[i];
        dv[i] = accumulate ? dv[i] + t : t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dg[0] = dscale[0] / nrm;
}
__global__ __launch_bounds__(256) void dot_kernel(const float* a, const float* b, long long n, double* acc) {
    double s = 0.0;
    GS_LOOP(i, n) s += (double)a[i] * (double)b[i];
    s = wave_sum_f64(s);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) acc[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void scale_finish_kernel(float* G, const float* scale_dev, long long n,
                                                           const double* part, int nparts, float* dot_out) {
    const float s = scale_dev[0];
    GS_LOOP(i, n) G[i] *= s;
    if (blockIdx.x == 0) {
        const double tot = sum_partials_256(part, nparts);
        if (threadIdx.x == 0) dot_out[0] = (float)tot;
    }
}

// ---- weight norm of MANY layers at once (the weights do not change inside a step: all norms in two launches at its start)
constexpr int WN_MULTI_MAX = 32, WN_MULTI_NB = 128;
struct WnMulti { const float* v[WN_MULTI_MAX]; const float* g[WN_MULTI_MAX]; long long n[WN_MULTI_MAX]; int count; };
__global__ __launch_bounds__(256) void wn_multi_sumsq_kernel(WnMulti a, double* part /* [count][NB] */) {
    const int t = blockIdx.y, j = blockIdx.x;
    const float* v = a.v[t];
    const long long n = a.n[t];
    double s = 0.0;
    if ((reinterpret_cast<uintptr_t>(v) & 15) == 0) {          // 16-byte loads, four independent accumulators in flight
        const long long n4 = n >> 2;
        const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
        for (long long i = (long long)j * 256 + threadIdx.x; i < n4; i += (long long)WN_MULTI_NB * 256) {
            const f32x4 x = v4[i];
            s += ((double)x[0] * x[0] + (double)x[1] * x[1]) + ((double)x[2] * x[2] + (double)x[3] * x[3]);
        }
        if (j == 0 && threadIdx.x < (n & 3)) { const double x = v[4 * n4 + threadIdx.x]; s += x * x; }
    } else {
        for (long long i = (long long)j * 256 + threadIdx.x; i < n; i += (long long)WN_MULTI_NB * 256) { const double x = v[i]; s += x * x; }
    }
    s = wave_sum_f64(s);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[t * WN_MULTI_NB + j] = (w[0] + w[1]) + (w[2] + w[3]);
}
__global__ __launch_bounds__(64) void wn_multi_finish_kernel(WnMulti a, const double* part, float* scales, float* norms) {
    const int t = blockIdx.x, l = threadIdx.x;
    double s = 0.0;
    for (int q = l; q < WN_MULTI_NB; q += 64) s += part[t * WN_MULTI_NB + q];
    s = wave_sum_f64(s);
    if (l == 0) {
        const float nrm = (float)sqrt(s);
        norms[t] = nrm;
        scales[t] = a.g[t][0] / nrm;
    }
}
// ---- backward through the weight norm of MANY layers at once (the raw weight gradients of a step are all there at the end of backward):
// part[t][j] = partial <G_t, v_t> (f64), then every workgroup of layer t re-sums the layer's WN_MULTI_NB partials in a fixed order and
// rewrites its slice  G = scale * G - (<G, v> g / ||v||^3) v ;  dg_t = <G, v> / ||v||.   Two launches instead of two per layer.
struct WnGradMulti { float* G[WN_MULTI_MAX]; const float* v[WN_MULTI_MAX]; const float* g[WN_MULTI_MAX]; const float* norm[WN_MULTI_MAX];
                     const float* scale[WN_MULTI_MAX]; float* dg[WN_MULTI_MAX]; long long n[WN_MULTI_MAX]; int count; };
__global__ __launch_bounds__(256) void wn_grad_multi_dot_kernel(WnGradMulti a, double* part /* [count][NB] */) {
    const int t = blockIdx.y, j = blockIdx.x;
    const float* G = a.G[t];
    const float* v = a.v[t];
    const long long n = a.n[t];
    double s = 0.0;
    if (((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(G)) & 15) == 0) {
        const long long n4 = n >> 2;
        const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(G);
        for (long long i = (long long)j * 256 + threadIdx.x; i < n4; i += (long long)WN_MULTI_NB * 256) {
            const f32x4 x = v4[i], y = g4[i];
            s += ((double)x[0] * y[0] + (double)x[1] * y[1]) + ((double)x[2] * y[2] + (double)x[3] * y[3]);
        }
        if (j == 0 && threadIdx.x < (n & 3)) s += (double)v[4 * n4 + threadIdx.x] * (double)G[4 * n4 + threadIdx.x];
    } else {
        for (long long i = (long long)j * 256 + threadIdx.x; i < n; i += (long long)WN_MULTI_NB * 256) s += (double)v[i] * (double)G[i];
    }
    s = wave_sum_f64(s);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[t * WN_MULTI_NB + j] = (w[0] + w[1]) + (w[2] + w[3]);
}
__global__ __launch_bounds__(256) void wn_grad_multi_finish_kernel(WnGradMulti a, const double* part) {
    const int t = blockIdx.y;
    __shared__ double w[4];
    __shared__ float bc;
    {
        double s = 0.0;                       // the layer's partials in a fixed order: lanes stride them, waves / lanes combined as below
        for (int q = threadIdx.x; q < WN_MULTI_NB; q += 256) s += part[t * WN_MULTI_NB + q];
        s = wave_sum_f64(s);
        if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) bc = (float)((w[0] + w[1]) + (w[2] + w[3]));
        __syncthreads();
    }
    const float dscale = bc, nrm = a.norm[t][0], sc = a.scale[t][0];
    const float coef = -dscale * a.g[t][0] / (nrm * nrm * nrm);
    float* G = a.G[t];
    const float* v = a.v[t];
    const long long n = a.n[t];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) G[i] = sc * G[i] + coef * v[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.dg[t][0] = dscale / nrm;
}

// ---- fused backward of one weight-normed layer: G (unscaled wgrad, in place) -> dv = scale * G - (<G, v> g / ||v||^3) v;
// dg = <G, v> / ||v||.  Every workgroup re-reads the <= 256 dot partials (fixed order: deterministic).
__global__ __launch_bounds__(256) void wn_grad_finish_kernel(float* G, const float* __restrict__ v, const float* g, const float* norm,
                                                             const float* scale, long long n, const double* part, int nparts, float* dg) {
    const double tot = sum_partials_256(part, nparts);
    __shared__ float bc[1];
    if (threadIdx.x == 0) bc[0] = (float)tot;
    __syncthreads();
    const float dscale = bc[0];
    const float nrm = norm[0], sc = scale[0];
    const float coef = -dscale * g[0] / (nrm * nrm * nrm);
    GS_LOOP(i, n) G[i] = sc * G[i] + coef * v[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) dg[0] = dscale / nrm;
}

// ------------------------------------------------------------------ reductions
// column sums of [M][N]: one thread per column (coalesced rows), four independent accumulators so four row loads are
// in flight per lane; the grid is cut along rows so that ~1024 workgroups stream concurrently; partial sums by atomics
// PART = false: one row block, plain store.  PART = true: row block y writes its partial sums to partial[y][N]; colsum_finish_kernel
// adds the partials in block order (fixed order: the result does not depend on scheduling -- float atomics made two runs of
// the same step differ in the last bits of every bias gradient, which Adam amplifies into visibly different weights).
template <bool PART>
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int M, int N, int ldx, float* out, int rows_per_blk) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_blk;
    const int r1 = min(M, r0 + rows_per_blk);
    const float* p = x + (long long)r0 * ldx + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += p[0]; s1 += p[ldx]; s2 += p[2 * (long long)ldx]; s3 += p[3 * (long long)ldx];
        p += 4 * (long long)ldx;
    }
    for (; r < r1; ++r) { s0 += p[0]; p += ldx; }
    if (PART) out[(long long)blockIdx.y * N + n] = (s0 + s1) + (s2 + s3);
    else out[n] = (s0 + s1) + (s2 + s3);
}
// 64 columns x 4 partial lanes per workgroup: lane q adds partials q, q+4, ... (eight loads in flight), the four lane sums are
// combined in lane order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nparts, int N, float* out, int accumulate) {
    __shared__ float sm[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int p = q;
        for (; p + 28 < nparts; p += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(p + 4 * u) * N + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) { if (u & 1) s1 += v[u]; else s0 += v[u]; }
        }
        for (; p < nparts; p += 4) s0 += partial[(long long)p * N + n];
    }
    sm[q][c] = s0 + s1;
    __syncthreads();
    if (q == 0 && n < N) {
        const float v = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
        out[n] = accumulate ? out[n] + v : v;
    }
}
__global__ __launch_bounds__(256) void rowsum_short_kernel(const float* x, long long rows, int cols, float* out, int as_mask) {
    GS_LOOP(r, rows) {                          // short rows (one-hot label channels): one thread per row
        const float* p = x + r * cols;
        float s = 0.f;
        for (int c = 0; c < cols; ++c) s += p[c];
        out[r] = as_mask ? (s != 0.f ? 1.f : 0.f) : s;
    }
}
__global__ __launch_bounds__(256) void rowsum_kernel(const float* x, long long rows, int cols, float* out, int as_mask) {
    // one wave per row
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* p = x + r * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += p[c];
        s = wave_sum(s);
        if (lane == 0) out[r] = as_mask ? (s != 0.f ? 1.f : 0.f) : s;
    }
}

// ------------------------------------------------------------------ element-wise
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, const float* y, float* dz, long long n, int act) {
    GS_LOOP(i, n) dz[i] = dy[i] * act_grad_from_out(y[i], act);
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, float* y, long long n, float p, float inv_keep,
                                                      DropSeed seed) {
    GS_LOOP(i, n) y[i] = x[i] * dropout_scale(seed, (uint64_t)i, p, inv_keep);
}
// T independent masks over the same index range: step t uses seed + t; x advances by x_step elements per step (0: every
// step reads the same input -- the decoder's per-step dropout of one embedding)
__global__ __launch_bounds__(256) void dropout_steps_kernel(const float* x, float* y, long long n, int steps, long long x_step,
                                                            float p, float inv_keep, DropSeed seed) {
    const long long total = n * steps;
    GS_LOOP(j, total) {
        const int t = (int)(j / n);
        const long long i = j - (long long)t * n;
        y[j] = x[t * x_step + i] * dropout_scale(DropSeed{seed.s + (uint64_t)t, seed.ep}, (uint64_t)i, p, inv_keep);
    }
}
struct SumArgs { const float* x[8]; int k; };
__global__ __launch_bounds__(256) void sum_n_kernel(SumArgs a, float* out, long long n4) {
    GS_LOOP(i, n4) {
        f32x4 s = reinterpret_cast<const f32x4*>(a.x[0])[i];
#pragma unroll
        for (int j = 1; j < 8; ++j)
            if (j < a.k) s += reinterpret_cast<const f32x4*>(a.x[j])[i];
        reinterpret_cast<f32x4*>(out)[i] = s;
    }
}
__global__ __launch_bounds__(256) void sum_n_tail_kernel(SumArgs a, float* out, long long from, long long n) {
    GS_LOOP(i0, n - from) {
        const long long i = from + i0;
        float s = a.x[0][i];
        for (int j = 1; j < a.k; ++j) s += a.x[j][i];
        out[i] = s;
    }
}
__global__ __launch_bounds__(256) void axpby_kernel(const float* x, const float* y, float* out, long long n, float a, float b) {
    GS_LOOP(i, n) out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}
// (a*x + b*x) + c*x in that order: what two axpby launches wrote (quirk Q-A, SelfMix3Fn)
__global__ __launch_bounds__(256) void self_mix3_kernel(const float* x, float* out, long long n, float a, float b, float c) {
    GS_LOOP(i, n) {
        const float v = x[i];
        const float t = a * v + b * v;
        out[i] = 1.0f * t + c * v;
    }
}
// zero-row mask of x (rowsum_kernel's summation order) and a dropped copy of x in one pass: the question-expansion mask and the
// input dropout of the FCNet that reads x next (relation_encoder.py:19-29, fc.py:36-37)
__global__ __launch_bounds__(256) void rowmask_dropout_kernel(const float* x, long long rows, int cols, float* mask, float* y, float p,
                                                              float inv_keep, DropSeed seed) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* q = x + r * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float v = q[c];
            s += v;
            y[r * cols + c] = v * dropout_scale(seed, (uint64_t)(r * cols + c), p, inv_keep);
        }
        s = wave_sum(s);
        if (lane == 0) mask[r] = s != 0.f ? 1.f : 0.f;
    }
}
__global__ __launch_bounds__(256) void mul_kernel(const float* x, const float* y, float* out, long long n) {
    GS_LOOP(i, n) out[i] = x[i] * y[i];
}
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* q, const float* mask, long long rows, int group,
                                                          int qrows, int cols, float* out, float p, float inv_keep, DropSeed seed) {
    const long long n = rows * cols;
    GS_LOOP(i, n) {
        const long long r = i / cols;
        const int c = (int)(i % cols);
        out[i] = mask[r] * q[((r / group) % qrows) * cols + c] * dropout_scale(seed, (uint64_t)i, p, inv_keep);
    }
}
// the same for cols % 4 == 0, one workgroup per output row: no per-element 64-bit divisions (three of them made the kernel above
// ALU-bound: 57 us for 75 MB), 16-byte accesses
__global__ __launch_bounds__(256) void expand_rows_vec_kernel(const float* __restrict__ q, const float* __restrict__ mask, int group, int qrows,
                                                              int cols, float* __restrict__ out, float p, float inv_keep, DropSeed seed) {
    const long long r = blockIdx.x;
    const float m = mask[r];
    const float* qr = q + (long long)((r / group) % qrows) * cols;
    for (int c = 4 * threadIdx.x; c < cols; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qr + c);
        const uint64_t i = (uint64_t)(r * cols + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = m * v[e] * dropout_scale(seed, i + e, p, inv_keep);
        *reinterpret_cast<f32x4*>(out + r * cols + c) = o;
    }
}
__global__ __launch_bounds__(256) void expand_rows_bwd_kernel(const float* dout, const float* mask, long long rows,
                                                              int group, int qrows, int cols, float* dq, float p, float inv_keep,
                                                              DropSeed seed) {
    const long long G = rows / group;            // row groups; group g reads q row (g % qrows)
    const long long n = (long long)qrows * cols;
    GS_LOOP(i, n) {
        const long long qr = i / cols;
        const int c = (int)(i % cols);
        float s = 0.f;
        for (long long g = qr; g < G; g += qrows)
            for (int k = 0; k < group; ++k) {
                const long long r = g * group + k;
                s += mask[r] * dout[r * cols + c] * dropout_scale(seed, (uint64_t)(r * cols + c), p, inv_keep);
            }
        dq[i] = s;
    }
}
__global__ __launch_bounds__(256) void residual_relu2_fwd_kernel(const float* v, const float* o, float* vnew, long long n,
                                                                 float p, float inv_keep, DropSeed seed) {
    GS_LOOP(i, n) {
        const float t = (o[i] + o[i]) * dropout_scale(seed, (uint64_t)i, p, inv_keep);
        vnew[i] = v[i] + (t > 0.f ? t : 0.f);
    }
}
__global__ __launch_bounds__(256) void residual_relu2_bwd_kernel(const float* o, const float* dvnew, float* d_o, long long n,
                                                                 float p, float inv_keep, DropSeed seed) {
    GS_LOOP(i, n) {
        const float k = dropout_scale(seed, (uint64_t)i, p, inv_keep);
        const float t = (o[i] + o[i]) * k;
        d_o[i] = t > 0.f ? dvnew[i] * 2.f * k : 0.f;
    }
}
// out = drop(sigmoid(g2 + g1)) * drop(tanh(c2 + c1)); c1/g1 (n1 elements: the terms shared by both images) broadcast over
// n = rep * n1 elements of c2/g2
__global__ __launch_bounds__(256) void gate_fuse_fwd_kernel(const float* c2, const float* g2, const float* c1, const float* g1,
                                                            float* out, long long n, long long n1, float p, float inv_keep,
                                                            DropSeed seed_c, DropSeed seed_g) {
    GS_LOOP(i, n) {
        long long j = i;
        if (n <= 4 * n1) { while (j >= n1) j -= n1; } else j = i % n1;       // (two images: at most one subtraction instead of a 64-bit division)
        const float c = tanhf(c2[i] + c1[j]) * dropout_scale(seed_c, (uint64_t)i, p, inv_keep);
        const float g = (1.f / (1.f + expf(-(g2[i] + g1[j])))) * dropout_scale(seed_g, (uint64_t)i, p, inv_keep);
        out[i] = g * c;
    }
}
__global__ __launch_bounds__(256) void gate_fuse_bwd_kernel(const float* c2, const float* g2, const float* c1, const float* g1,
                                                            const float* dout, float* dc2, float* dg2, float* dc1, float* dg1,
                                                            long long n, long long n1, float p, float inv_keep, DropSeed seed_c,
                                                            DropSeed seed_g) {
    GS_LOOP(j, n1) {
        float sc = 0.f, sg = 0.f;
        for (long long i = j; i < n; i += n1) {
            const float kc = dropout_scale(seed_c, (uint64_t)i, p, inv_keep);
            const float kg = dropout_scale(seed_g, (uint64_t)i, p, inv_keep);
            const float tc = tanhf(c2[i] + c1[j]);
            const float sgm = 1.f / (1.f + expf(-(g2[i] + g1[j])));
            const float d = dout[i];
            const float dc = d * (sgm * kg) * kc * (1.f - tc * tc);
            const float dg = d * (tc * kc) * kg * sgm * (1.f - sgm);
            dc2[i] = dc; dg2[i] = dg;
            sc += dc; sg += dg;
        }
        dc1[j] = sc; dg1[j] = sg;
    }
}

// ------------------------------------------------------------------ weighted pooling  out[g][c] = sum_i w[g][i] x[g][i][c]
__global__ __launch_bounds__(256) void weighted_pool_fwd_kernel(const float* w, const float* x, int G, int n, int C, float* out) {
    const long long total = (long long)G * C;
    GS_LOOP(i, total) {
        const long long g = i / C;
        const int c = (int)(i % C);
        const float* xp = x + g * n * C + c;
        const float* wp = w + g * n;
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += wp[k] * xp[(long long)k * C];
        out[i] = s;
    }
}
// dx[g][i][c] = w[g][i] * dout[g][c];  dw[g][i] = sum_c x[g][i][c] * dout[g][c]   (one wave per (g,i))
__global__ __launch_bounds__(256) void weighted_pool_bwd_kernel(const float* w, const float* x, const float* dout, int G, int n,
                                                                int C, float* dw, float* dx) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long rows = (long long)G * n;
    for (long long r = wave0; r < rows; r += nwaves) {
        const long long g = r / n;
        const float wv = w[r];
        const float* xp = x + r * C;
        const float* dp = dout + g * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = dp[c];
            s += xp[c] * d;
            dx[r * C + c] = wv * d;
        }
        s = wave_sum(s);
        if (lane == 0) dw[r] = s;
    }
}

// ------------------------------------------------------------------ question attention weights (quirk Q-C)
// S[l][b] = softmax over b of atten[b][l];  weight[j] = S_flat[j]  (weight viewed (B,L), S viewed (L,B))
__global__ __launch_bounds__(64) void qatt_weights_fwd_kernel(const float* atten, int B, int L, float* S, float* weight) {
    const int l = blockIdx.x, lane = threadIdx.x;
    float mx = -INFINITY;
    for (int b = lane; b < B; b += 64) mx = fmaxf(mx, atten[(long long)b * L + l]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int b = lane; b < B; b += 64) s += expf(atten[(long long)b * L + l] - mx);
    s = wave_sum(s);
    for (int b = lane; b < B; b += 64) {
        const float v = expf(atten[(long long)b * L + l] - mx) / s;
        S[(long long)l * B + b] = v;
        weight[(long long)l * B + b] = v;      // same flat memory order: the (B,L) view re-reads it
    }
}
__global__ __launch_bounds__(64) void qatt_weights_bwd_kernel(const float* S, const float* dweight, int B, int L, float* datten) {
    const int l = blockIdx.x, lane = threadIdx.x;
    float dot = 0.f;
    for (int b = lane; b < B; b += 64) dot += S[(long long)l * B + b] * dweight[(long long)l * B + b];
    dot = wave_sum(dot);
    for (int b = lane; b < B; b += 64) {
        const float sv = S[(long long)l * B + b];
        datten[(long long)b * L + l] = sv * (dweight[(long long)l * B + b] - dot);
    }
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* tok, long long ntok, const float* table, int E,
                                                            float* out, int ld_out, int col_off, int relu) {
    const long long n = ntok * E;
    GS_LOOP(i, n) {
        const long long t = i / E;
        const int e = (int)(i % E);
        float v = table[tok[t] * E + e];
        if (relu) v = v > 0.f ? v : 0.f;
        out[t * ld_out + col_off + e] = v;
    }
}
// dtable[id] += sum over the positions t with tok[t] == id, in position order.  One WAVE per (table row, 64 columns): the
// wave reads 64 token ids at a time (next chunk's load in flight), ballots the matches and walks the set bits in order --
// no LDS, no barrier, no float atomics; lane = column.
__global__ __launch_bounds__(64) void embedding_bwd_kernel(const int64_t* tok, long long ntok, const float* table,
                                                           const float* dout, int E, int ld_out, int col_off, int relu,
                                                           long long padding_idx, float* dtable) {
    constexpr int U = 16;                            // 16 x 64 token ids per round: sixteen independent loads per lane
    const int64_t id = blockIdx.x;
    if (id == padding_idx) return;
    const int lane = threadIdx.x;
    const int e = blockIdx.y * 64 + lane;
    const bool col = e < E;
    const bool live = col && (!relu || table[id * E + e] > 0.f);
    const float* dcol = dout + col_off + (col ? e : 0);
    float acc = 0.f;
    for (long long sbase = 0; sbase < ntok; sbase += 64 * U) {
        bool hit[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long t = sbase + 64 * u + lane;
            hit[u] = t < ntok && tok[t] == id;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned long long m = __ballot(hit[u]);                 // uniform over the wave
            const float* rowbase = dcol + (sbase + 64 * u) * ld_out;
            while (m) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {                        // up to four row loads in flight, added in position order
                    if (m) {
                        const int bpos = __builtin_ctzll(m);
                        m &= m - 1;
                        v[q] = rowbase[(long long)bpos * ld_out];
                    }
                }
                acc += v[0]; acc += v[1]; acc += v[2]; acc += v[3];
            }
        }
    }
    if (live && acc != 0.f) dtable[id * E + e] += acc;
}

// one-hot rows of the token ids (padding rows all zero): the A operand of  dtable = OneHot^T . dout  (a deterministic
// split-K GEMM whose cost does not depend on how skewed the id distribution is -- padding / '<start>' ids hit thousands of
// positions, which a per-row scan serialises)
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* tok, long long ntok, int V, long long padding_idx, float* out) {
    const long long n = ntok * V;
    GS_LOOP(i, n) {
        const long long t = i / V;
        const int v = (int)(i % V);
        const int64_t id = tok[t];
        out[i] = (id == v && id != padding_idx) ? 1.f : 0.f;
    }
}
// dtable += raw * (relu ? table > 0 : 1)
__global__ __launch_bounds__(256) void masked_add_kernel(const float* raw, const float* table, int relu, float* dtable, long long n) {
    GS_LOOP(i, n) {
        if (!relu || table[i] > 0.f) dtable[i] += raw[i];
    }
}

// ------------------------------------------------------------------ row softmax / log-softmax (one wave per row)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* x, float* y, long long rows, int cols, int log_mode) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* xp = x + r * cols;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xp[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += expf(xp[c] - mx);
        s = wave_sum(s);
        const float ls = logf(s);
        for (int c = lane; c < cols; c += 64) {
            const float t = xp[c] - mx;
            y[r * cols + c] = log_mode ? t - ls : expf(t) / s;
        }
    }
}
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* y, const float* dy, float* dx, long long rows, int cols,
                                                          int log_mode) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* yp = y + r * cols;
        const float* dp = dy + r * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += log_mode ? dp[c] : dp[c] * yp[c];
        s = wave_sum(s);
        for (int c = lane; c < cols; c += 64)
            dx[r * cols + c] = log_mode ? dp[c] - expf(yp[c]) * s : yp[c] * (dp[c] - s);
    }
}

// ------------------------------------------------------------------ mix3
__global__ __launch_bounds__(256) void mix3_fwd_kernel(const float* w, const float* a, const float* b, const float* c, int B,
                                                       int C, float* out) {
    const long long n = (long long)B * C;
    GS_LOOP(i, n) {
        const long long r = i / C;
        out[i] = w[r * 3] * a[i] + w[r * 3 + 1] * b[i] + w[r * 3 + 2] * c[i];
    }
}
__global__ __launch_bounds__(256) void mix3_bwd_kernel(const float* w, const float* a, const float* b, const float* c,
                                                       const float* dout, int B, int C, float* dw, float* da, float* db,
                                                       float* dc, int accumulate) {
    // one wave per row
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < B; r += nwaves) {
        const float w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int k = lane; k < C; k += 64) {
            const long long i = r * C + k;
            const float d = dout[i];
            s0 += d * a[i]; s1 += d * b[i]; s2 += d * c[i];
            if (accumulate) { da[i] += w0 * d; db[i] += w1 * d; dc[i] += w2 * d; }
            else { da[i] = w0 * d; db[i] = w1 * d; dc[i] = w2 * d; }
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { dw[r * 3] = s0; dw[r * 3 + 1] = s1; dw[r * 3 + 2] = s2; }
    }
}

// ------------------------------------------------------------------ masked NLL
// one workgroup (B * T <= a few ten thousand terms): strided partial sums, fixed-order tree -- deterministic; the division by
// the (optionally external) denominator is folded in
__global__ __launch_bounds__(1024) void nll_fwd_kernel(const float* logp, const int64_t* target, int ld_t, const float* mask,
                                                       int ld_m, int B, int T, int V, const float* denom, float* sums, float* loss) {
    __shared__ float ss[1024], sm[1024];
    float s = 0.f, ms = 0.f;
    const long long n = (long long)B * T;
    for (long long i0 = threadIdx.x; i0 < n; i0 += 4 * 1024) {          // four independent gathers in flight per thread
        float lp[4], mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + u * 1024;
            const bool ok = i < n;
            const long long b = ok ? i / T : 0;
            const int t = ok ? (int)(i % T) : 0;
            mk[u] = ok ? mask[b * ld_m + t] : 0.f;
            lp[u] = logp[(b * T + t) * V + target[b * ld_t + t]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s -= lp[u] * mk[u]; ms += mk[u]; }
    }
    ss[threadIdx.x] = s; sm[threadIdx.x] = ms;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { ss[threadIdx.x] += ss[threadIdx.x + w]; sm[threadIdx.x] += sm[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = ss[0];
        sums[1] = denom ? denom[0] : sm[0];
        loss[0] = sums[0] / sums[1];
    }
}
__global__ __launch_bounds__(256) void nll_bwd_kernel(const int64_t* target, int ld_t, const float* mask, int ld_m,
                                                      const float* sums, const float* gscale, int B, int T, int V, float* dlogp) {
    const long long n = (long long)B * T * V;
    const float sc = -gscale[0] / sums[1];
    GS_LOOP(i, n) {
        const int v = (int)(i % V);
        const long long bt = i / V;
        const long long b = bt / T;
        const int t = (int)(bt % T);
        dlogp[i] = (target[b * ld_t + t] == v) ? sc * mask[b * ld_m + t] : 0.f;
    }
}

// denom (optional): replaces the local sum of the mask -- the global-batch mask sum / world size under data parallelism,
// so that the average of the per-rank losses is the reference's loss over the global batch (utils/utils.py:213-214)
__global__ __launch_bounds__(256) void total_loss_kernel(const float* sl, const float* a, const float* b, long long n, float coef,
                                                         float* out) {
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) s += a[i] + b[i];
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sl[0] + coef * (part[0] + part[1] + part[2] + part[3]);
}
__global__ __launch_bounds__(256) void fill_scaled_kernel(const float* scalar, float coef, float* out, long long n) {
    const float v = scalar[0] * coef;
    GS_LOOP(i, n) out[i] = v;
}

// ------------------------------------------------------------------ LayerNorm (one wave per row)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, const float* gamma, const float* beta, long long rows,
                                                            int cols, float eps, float* y, float* save) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float* xp = x + r * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += xp[c];
        const float mean = wave_sum(s) / cols;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) { const float d = xp[c] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / cols + eps);
        for (int c = lane; c < cols; c += 64) y[r * cols + c] = (xp[c] - mean) * rstd * gamma[c] + beta[c];
        if (lane == 0) { save[r * 2] = mean; save[r * 2 + 1] = rstd; }
    }
}
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* x, const float* gamma, const float* save, const float* dy,
                                                            long long rows, int cols, float* dx, float* dgamma, float* dbeta) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < rows; r += nwaves) {
        const float mean = save[r * 2], rstd = save[r * 2 + 1];
        const float* xp = x + r * cols;
        const float* dp = dy + r * cols;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xp[c] - mean) * rstd;
            const float g = dp[c] * gamma[c];
            s1 += g; s2 += g * xh;
        }
        s1 = wave_sum(s1) / cols; s2 = wave_sum(s2) / cols;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xp[c] - mean) * rstd;
            dx[r * cols + c] = rstd * (dp[c] * gamma[c] - s1 - xh * s2);
        }
    }
}

// dgamma[c] = sum_r dy[r][c] * xhat[r][c], dbeta[c] = sum_r dy[r][c]: row-block partials (thread = column) for the fixed-order
// finish of the column sums (no float atomics)
__global__ __launch_bounds__(256) void layernorm_param_partial_kernel(const float* x, const float* save, const float* dy, long long rows,
                                                                      int cols, int rows_per_blk, float* pg, float* pb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long r0 = (long long)blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    float g = 0.f, b = 0.f;
    for (long long r = r0; r < r1; ++r) {
        const float d = dy[r * cols + c];
        g += d * (x[r * cols + c] - save[r * 2]) * save[r * 2 + 1];
        b += d;
    }
    pg[(long long)blockIdx.y * cols + c] = g;
    pb[(long long)blockIdx.y * cols + c] = b;
}

// ------------------------------------------------------------------ label matrix -> one-hot adjacency
// ref: utils/mimic_utils.py:119-149.  adj: [B][S][S] (float64 as collated, or int64), cropped to NxN.
template <typename T>
__global__ __launch_bounds__(256) void process_matrix_kernel(const T* adj, int B, int S, int N, int L, float* out) {
    const long long n = (long long)B * N * N * L;
    GS_LOOP(i, n) {
        const int l = (int)(i % L);
        long long rest = i / L;
        const int j = (int)(rest % N); rest /= N;
        const int ii = (int)(rest % N);
        const long long b = rest / N;
        const T v = adj[(b * S + ii) * S + j];
        out[i] = (v == (T)(l + 1)) ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_bc1,
                                                   float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
    GS_LOOP(i, n) {
        const float gr = g[i] * gscale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gr;
        const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// Adam whose step counter and hyper-parameters live on the DEVICE (graph-capturable: a captured launch of the by-value
// form would freeze the bias correction and the learning rate).  state[0] = step (incremented here, as a float-exact
// integer < 2^24 steps... stored as int), hyper = {lr, beta1, beta2, eps, grad_scale}.
// A persistent sequence kernel that abandoned its launch (csrc/persist*.hip: a seam timed out) leaves garbage gradients behind.
// The update is then SKIPPED on the device -- step counter, moments and weights untouched -- when the local status word is set or
// when `fault` (a flag every rank adds into the all-reduced gradient buffer, ekaid_seq_fault_flag) is non-zero on any rank.
__global__ void adam_prep_kernel(int* step, const float* hyper, float* derived, const unsigned* status, const float* fault) {
    if ((status && status[0] != 0u) || (fault && fault[0] != 0.f)) { derived[2] = 1.f; return; }
    derived[2] = 0.f;
    const int t = step[0] + 1;
    step[0] = t;
    const double b1 = hyper[1], b2 = hyper[2];
    derived[0] = (float)((double)hyper[0] / (1.0 - pow(b1, (double)t)));
    derived[1] = (float)(1.0 / sqrt(1.0 - pow(b2, (double)t)));
}
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                                                       const float* derived) {
    if (derived[2] != 0.f) return;            // (uniform over the grid: the whole update is skipped)
    const float lr_bc1 = derived[0], inv_sqrt_bc2 = derived[1], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], gscale = hyper[4];
    GS_LOOP(i, n) {
        const float gr = g[i] * gscale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gr;
        const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// scratch for the two-stage weight-norm reductions: per-workgroup partials of sumsq / dot (two slots of 2048 doubles),
// summed in a fixed order by the finish kernels (deterministic, no atomics, no zero-fill), followed by the row-block
// partials of the column sums.  One buffer per DEVICE (part of the per-device state, api.hip), shared by the stream-ordered
// calls of one stream: ekaid_wn_scale_fwd, ekaid_scale_dot and ekaid_colsum(_acc) must not run concurrently on two streams
// of the same device.  Allocated by ekaid_init / at first use
// (outside any graph capture: ekaid_amd.hip warms it up).
constexpr int RED_SLOT = 2048;
constexpr long long COLSUM_PART_FLOATS = 1 << 18;       // column-sum partials [row blocks][N] behind the two reduction slots (1 MiB)
static double* reduction_scratch() { return device_scratch(EKAID_SCRATCH_BYTES); }

}  // namespace ekaid

using namespace ekaid;
#define ST ((hipStream_t)stream)
#define INVK(p) ((p) > 0.f ? 1.f / (1.f - (p)) : 1.f)

extern "C" int ekaid_wn_scale_fwd(const float* v, const float* g, long long n, float* scale_out, float* norm_out, void* stream) {
    double* acc = reduction_scratch();
    if (!acc) { set_error("wn_scale_fwd: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    const dim3 grid = grid1d(n, 4);               // <= 2048 workgroups = RED_SLOT partials
    hipLaunchKernelGGL(sumsq_kernel, grid, dim3(256), 0, ST, v, n, acc);
    hipLaunchKernelGGL(wn_finish_kernel, dim3(1), dim3(256), 0, ST, (const double*)acc, (int)grid.x, g, scale_out, norm_out);
    return check_launch("wn_scale_fwd");
}
extern "C" int ekaid_wn_scale_bwd(const float* v, const float* g, const float* norm, const float* dscale, long long n,
                                  float* dv, int accumulate_dv, float* dg, void* stream) {
    hipLaunchKernelGGL(wn_bwd_kernel, grid1d(n), dim3(256), 0, ST, v, g, norm, dscale, n, dv, accumulate_dv, dg);
    return check_launch("wn_scale_bwd");
}
extern "C" int ekaid_scale_dot(float* G, const float* w, const float* scale_dev, long long n, float* dot_out, void* stream) {
    double* acc = reduction_scratch();
    if (!acc) { set_error("scale_dot: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    const dim3 grid = grid1d(n, 4);
    hipLaunchKernelGGL(dot_kernel, grid, dim3(256), 0, ST, (const float*)G, w, n, acc + RED_SLOT);
    hipLaunchKernelGGL(scale_finish_kernel, grid1d(n), dim3(256), 0, ST, G, scale_dev, n, (const double*)(acc + RED_SLOT), (int)grid.x,
                       dot_out);
    return check_launch("scale_dot");
}
extern "C" int ekaid_wn_scale_multi(const float* const* v, const float* const* g, const long long* n, int count, float* scales,
                                    float* norms, void* stream) {
    if (count < 1 || count > WN_MULTI_MAX || !v || !g || !n || !scales || !norms) {
        set_error("wn_scale_multi: 1 <= count <= %d layers, non-null arrays", WN_MULTI_MAX);
        return EKAID_ERR_ARG;
    }
    double* acc = reduction_scratch();
    if (!acc) { set_error("wn_scale_multi: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    WnMulti a;
    a.count = count;
    for (int i = 0; i < WN_MULTI_MAX; ++i) {
        a.v[i] = i < count ? v[i] : nullptr; a.g[i] = i < count ? g[i] : nullptr; a.n[i] = i < count ? n[i] : 0;
        if (i < count && (!v[i] || !g[i] || n[i] < 1)) { set_error("wn_scale_multi: null / empty layer %d", i); return EKAID_ERR_ARG; }
    }
    double* part = acc + 2 * RED_SLOT;          // the column-sum partial area (1 MiB): 32 x 128 doubles = 32 KiB
    hipLaunchKernelGGL(wn_multi_sumsq_kernel, dim3(WN_MULTI_NB, count), dim3(256), 0, ST, a, part);
    hipLaunchKernelGGL(wn_multi_finish_kernel, dim3(count), dim3(64), 0, ST, a, (const double*)part, scales, norms);
    return check_launch("wn_scale_multi");
}
// ---- row and column maxima of |x| for up to AMAX_MAX row-strided matrices in one launch: the operand bounds of the two-term f16
// GEMM split (ekaid_gemm_t.a_amax / b_amax).  |x| as an unsigned bit pattern orders like the value and an integer max does not depend
// on arrival order: deterministic.  A NaN element has the largest pattern of all, so it surfaces in the bound (the consumer then
// leaves that row unscaled).
// Work split: a workgroup owns ONE 256-column slice (each lane one 16-byte vector per row) and every `panels`-th 64-row chunk of it;
// a wave takes 16 rows of the chunk, all 16 loads in flight.  Column maxima stay in four registers per lane for the workgroup's whole
// life, meet the other waves' in LDS at the end and leave as ONE atomic per column and workgroup -- so an output word sees `panels`
// atomics, not rows / 32 of them: same-address atomics retire one after the other (~40 ns each, measured: the 576-panel form of this
// kernel spent 23 us on a 9 MB operand), and `panels` <= 64 keeps that tail under ~3 us.  A row's maximum is reduced across the lanes
// (DPP), parked in the lane numbered like the row, and leaves as one 16-lane atomic per wave and chunk (a plain store when the matrix
// has a single slice): contention there is the slice count.
constexpr int AMAX_MAX = 32;
constexpr int AMAX_SLICE = 256;                           // columns per workgroup
constexpr int AMAX_CHUNK = 64;                            // rows per workgroup iteration: 16 per wave
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct AmaxMulti { const float* x[AMAX_MAX]; long long rows[AMAX_MAX]; int cols[AMAX_MAX]; int ld[AMAX_MAX]; unsigned* rowmax[AMAX_MAX]; unsigned* colmax[AMAX_MAX]; int count; };
// max over the 64 lanes of a wave, valid in lane 63 (max is idempotent: any exchange pattern that reaches every lane will do)
__device__ __forceinline__ unsigned wave_max_u32_lane63(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));     // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));     // row_mirror: every lane = its row of 16
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true));     // row_bcast15 into rows 1, 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true));     // row_bcast31 into rows 2, 3
    return v;
}
__global__ __launch_bounds__(256) void amax_rc_kernel(AmaxMulti a) {
    constexpr int RW = AMAX_CHUNK / 4;                    // rows per wave and chunk
    __shared__ unsigned cmax[4][AMAX_SLICE];
    const int t = blockIdx.z;
    const long long rows = a.rows[t];
    const int cols = a.cols[t], ld = a.ld[t];
    const int col0 = blockIdx.y * AMAX_SLICE;
    if (col0 >= cols || rows <= 0) return;
    const float* x = a.x[t];
    unsigned* const rowmax = a.rowmax[t];
    unsigned* const colmax = a.colmax[t];
    const bool one_slice = cols <= AMAX_SLICE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool ok = col0 + 4 * lane < cols;
    const long long nchunks = (rows + AMAX_CHUNK - 1) / AMAX_CHUNK;
    unsigned cm[4] = {0, 0, 0, 0};
    for (long long ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const long long rbase = ch * AMAX_CHUNK + wave * RW;
        if (rbase >= rows) continue;                                                  // (a ragged last chunk: this wave has no row in it)
        u32x4_t v[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            long long row = rbase + u;
            row = row < rows ? row : rows - 1;                                        // (clamped: a repeated row changes no column maximum;
            v[u] = ok ? *reinterpret_cast<const u32x4_t*>(x + row * ld + col0 + 4 * lane) : u32x4_t{0, 0, 0, 0};      // its row maximum is not stored)
        }
        unsigned mine = 0;
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned av = v[u][e] & 0x7FFFFFFFu;
                cm[e] = max(cm[e], av);
                m = max(m, av);
            }
            if (rowmax) {
                const unsigned whole = (unsigned)__builtin_amdgcn_readlane((int)wave_max_u32_lane63(m), 63);
                mine = lane == u ? whole : mine;
            }
        }
        if (rowmax && lane < RW && rbase + lane < rows) {
            if (one_slice) rowmax[rbase + lane] = mine;
            else if (mine) atomicMax(rowmax + rbase + lane, mine);
        }
    }
    if (colmax) {
        *reinterpret_cast<u32x4_t*>(&cmax[wave][4 * lane]) = u32x4_t{cm[0], cm[1], cm[2], cm[3]};
        __syncthreads();
        const unsigned m = max(max(cmax[0][tid], cmax[1][tid]), max(cmax[2][tid], cmax[3][tid]));
        if (m && col0 + tid < cols) atomicMax(colmax + col0 + tid, m);
    }
}
static int amax_panels_override() {
    static const int v = [] { const char* e = getenv("EKAID_AMAX_PANELS"); return e ? atoi(e) : 0; }();
    return v;
}
extern "C" int ekaid_amax_rc_multi(const float* const* x, const long long* rows, const int* cols, const int* ld, float* const* rowmax,
                                   float* const* colmax, int count, int outputs_zeroed, void* stream) {
    if (count < 1 || count > AMAX_MAX || !x || !rows || !cols || !ld || !rowmax || !colmax) {
        set_error("amax_rc_multi: 1 <= count <= %d matrices, non-null arrays", AMAX_MAX);
        return EKAID_ERR_ARG;
    }
    AmaxMulti a;
    a.count = count;
    long long rmax = 0;
    int cmaxw = 0;
    for (int i = 0; i < AMAX_MAX; ++i) {
        const bool on = i < count;
        if (on && (!x[i] || rows[i] < 0 || cols[i] < 0 || ld[i] < cols[i])) { set_error("amax_rc_multi: bad matrix %d", i); return EKAID_ERR_ARG; }
        if (on && ((cols[i] & 3) || (ld[i] & 3) || (reinterpret_cast<uintptr_t>(x[i]) & 15))) {
            set_error("amax_rc_multi: matrix %d is not made of whole 16-byte vectors (cols %d, ld %d)", i, cols[i], ld[i]);
            return EKAID_ERR_UNSUPPORTED;
        }
        a.x[i] = on ? x[i] : nullptr; a.rows[i] = on ? rows[i] : 0; a.cols[i] = on ? cols[i] : 0; a.ld[i] = on ? ld[i] : 0;
        a.rowmax[i] = on ? reinterpret_cast<unsigned*>(rowmax[i]) : nullptr; a.colmax[i] = on ? reinterpret_cast<unsigned*>(colmax[i]) : nullptr;
        if (on && rows[i] > 0 && cols[i] > 0) {
            if (rows[i] > rmax) rmax = rows[i];
            if (cols[i] > cmaxw) cmaxw = cols[i];
        }
        // both outputs are atomic-max targets (the row output only when the matrix has several slices; one without columns has no
        // workgroup at all): cleared here unless the caller hands in zeroed memory
        if (on && rowmax[i] && rows[i] > 0 && (cols[i] == 0 || (!outputs_zeroed && cols[i] > AMAX_SLICE)) &&
            hipMemsetAsync(rowmax[i], 0, sizeof(float) * rows[i], ST) != hipSuccess) { (void)hipGetLastError(); set_error("amax_rc_multi: memset failed"); return EKAID_ERR_LAUNCH; }
        if (on && colmax[i] && cols[i] > 0 && !outputs_zeroed &&
            hipMemsetAsync(colmax[i], 0, sizeof(float) * cols[i], ST) != hipSuccess) { (void)hipGetLastError(); set_error("amax_rc_multi: memset failed"); return EKAID_ERR_LAUNCH; }
    }
    if (rmax == 0) return EKAID_OK;
    // row panels: enough workgroups (panels x slices x matrices) to cover the chip about twice, never more than 64 per matrix (the
    // same-address atomic tail above) nor than there are 64-row chunks
    const int slices = (cmaxw + AMAX_SLICE - 1) / AMAX_SLICE;
    const long long nchunks = (rmax + AMAX_CHUNK - 1) / AMAX_CHUNK;
    long long panels = (512 + (long long)slices * count - 1) / ((long long)slices * count);
    panels = panels > 64 ? 64 : panels;
    if (amax_panels_override() > 0) panels = amax_panels_override();
    panels = panels > nchunks ? nchunks : panels;
    panels = panels < 1 ? 1 : panels;
    hipLaunchKernelGGL(amax_rc_kernel, dim3((unsigned)panels, (unsigned)slices, (unsigned)count), dim3(256), 0, ST, a);
    return check_launch("amax_rc_multi");
}
extern "C" int ekaid_amax_rc(const float* x, long long rows, int cols, int ld, float* rowmax, float* colmax, int outputs_zeroed, void* stream) {
    return ekaid_amax_rc_multi(&x, &rows, &cols, &ld, &rowmax, &colmax, 1, outputs_zeroed, stream);
}

extern "C" int ekaid_wn_grad_multi(float* const* G, const float* const* v, const float* const* g, const float* const* norm,
                                   const float* const* scale, const long long* n, float* const* dg, int count, void* stream) {
    if (count < 1 || count > WN_MULTI_MAX || !G || !v || !g || !norm || !scale || !n || !dg) {
        set_error("wn_grad_multi: 1 <= count <= %d layers, non-null arrays", WN_MULTI_MAX);
        return EKAID_ERR_ARG;
    }
    double* acc = reduction_scratch();
    if (!acc) { set_error("wn_grad_multi: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    WnGradMulti a;
    a.count = count;
    long long nmax = 0;
    for (int i = 0; i < WN_MULTI_MAX; ++i) {
        const bool on = i < count;
        if (on && (!G[i] || !v[i] || !g[i] || !norm[i] || !scale[i] || !dg[i] || n[i] < 1)) { set_error("wn_grad_multi: null / empty layer %d", i); return EKAID_ERR_ARG; }
        a.G[i] = on ? G[i] : nullptr; a.v[i] = on ? v[i] : nullptr; a.g[i] = on ? g[i] : nullptr; a.norm[i] = on ? norm[i] : nullptr;
        a.scale[i] = on ? scale[i] : nullptr; a.dg[i] = on ? dg[i] : nullptr; a.n[i] = on ? n[i] : 0;
        if (on && n[i] > nmax) nmax = n[i];
    }
    double* part = acc + 2 * RED_SLOT;          // the column-sum partial area (1 MiB): 32 x 128 doubles = 32 KiB
    hipLaunchKernelGGL(wn_grad_multi_dot_kernel, dim3(WN_MULTI_NB, count), dim3(256), 0, ST, a, part);
    int nb = (int)((nmax + 2047) / 2048);       // eight elements per thread of the largest layer; smaller layers leave idle workgroups
    nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
    hipLaunchKernelGGL(wn_grad_multi_finish_kernel, dim3(nb, count), dim3(256), 0, ST, a, (const double*)part);
    return check_launch("wn_grad_multi");
}
extern "C" int ekaid_wn_grad_fused(float* G, const float* v, const float* g, const float* norm, const float* scale, long long n,
                                   float* dg, void* stream) {
    if (!G || !v || !g || !norm || !scale || !dg) { set_error("wn_grad_fused: null pointer"); return EKAID_ERR_ARG; }
    if (n <= 0) return EKAID_OK;
    double* acc = reduction_scratch();
    if (!acc) { set_error("wn_grad_fused: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    dim3 grid = grid1d(n, 8);
    if (grid.x > 256) grid.x = 256;                 // <= 256 partials: every workgroup of the finish kernel re-sums them
    hipLaunchKernelGGL(dot_kernel, grid, dim3(256), 0, ST, (const float*)G, v, n, acc + RED_SLOT);
    hipLaunchKernelGGL(wn_grad_finish_kernel, grid1d(n), dim3(256), 0, ST, G, v, g, norm, scale, n, (const double*)(acc + RED_SLOT),
                       (int)grid.x, dg);
    return check_launch("wn_grad_fused");
}
// column sums: one row block -> plain store; several -> per-block partials in the per-device scratch + fixed-order finish
static int colsum_impl(const float* x, int M, int N, int ldx, float* out, int accumulate, hipStream_t st) {
    if (N <= 0) return EKAID_OK;
    if (M <= 0) { if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * N, st); return EKAID_OK; }
    const int colblks = cdiv(N, 256);
    int rows_per_blk = (int)(((long long)M * colblks + 511) / 512);          // ~512 workgroups stream the matrix
    rows_per_blk = rows_per_blk < 16 ? 16 : rows_per_blk;
    int nparts = cdiv(M, rows_per_blk);
    if (N > COLSUM_PART_FLOATS) {          // wider than the partial-sum scratch: column chunks, each a call of its own
        for (int c0 = 0; c0 < N; c0 += (int)COLSUM_PART_FLOATS) {
            const int rc = colsum_impl(x + c0, M, (int)(N - c0 < COLSUM_PART_FLOATS ? N - c0 : COLSUM_PART_FLOATS), ldx, out + c0, accumulate, st);
            if (rc) return rc;
        }
        return EKAID_OK;
    }
    while (nparts > 1 && (long long)nparts * N > COLSUM_PART_FLOATS) {      // (very wide N: fewer, longer row blocks)
        rows_per_blk = rows_per_blk > M / 2 ? M : rows_per_blk * 2;
        nparts = cdiv(M, rows_per_blk);
    }
    if (nparts == 1 && !accumulate) {
        hipLaunchKernelGGL((colsum_kernel<false>), dim3(colblks, 1), dim3(256), 0, st, x, M, N, ldx, out, rows_per_blk);
        return check_launch("colsum");
    }
    double* scr = reduction_scratch();
    if (!scr) { set_error("colsum: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    float* partial = reinterpret_cast<float*>(scr + 2 * RED_SLOT);
    hipLaunchKernelGGL((colsum_kernel<true>), dim3(colblks, nparts), dim3(256), 0, st, x, M, N, ldx, partial, rows_per_blk);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 64)), dim3(256), 0, st, (const float*)partial, nparts, N, out, accumulate);
    return check_launch("colsum");
}
extern "C" int ekaid_colsum(const float* x, int M, int N, int ldx, float* out, void* stream) { return colsum_impl(x, M, N, ldx, out, 0, ST); }
extern "C" int ekaid_colsum_acc(const float* x, int M, int N, int ldx, float* out, void* stream) { return colsum_impl(x, M, N, ldx, out, 1, ST); }
extern "C" int ekaid_sum_n(const float* const* xs, int k, float* out, long long n, void* stream) {
    if (k < 1 || k > 8 || !xs) { set_error("sum_n: 1 <= k <= 8 addends (got %d)", k); return EKAID_ERR_ARG; }
    if (n <= 0) return EKAID_OK;
    SumArgs a;
    bool vec = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int j = 0; j < 8; ++j) {
        a.x[j] = j < k ? xs[j] : nullptr;
        if (j < k && !xs[j]) { set_error("sum_n: null addend %d", j); return EKAID_ERR_ARG; }
        if (j < k && (reinterpret_cast<uintptr_t>(xs[j]) & 15)) vec = false;
    }
    a.k = k;
    const long long n4 = vec ? n / 4 : 0;
    if (n4) hipLaunchKernelGGL(sum_n_kernel, grid1d(n4), dim3(256), 0, ST, a, out, n4);
    if (4 * n4 < n) hipLaunchKernelGGL(sum_n_tail_kernel, grid1d(n - 4 * n4), dim3(256), 0, ST, a, out, 4 * n4, n);
    return check_launch("sum_n");
}
extern "C" int ekaid_dropout_steps(const float* x, float* y, long long n, int steps, long long x_step, float p, uint64_t seed,
                                   void* stream) {
    if (p < 0.f || p >= 1.f) { set_error("dropout_steps: p=%f out of range", p); return EKAID_ERR_ARG; }
    if (n <= 0 || steps <= 0) return EKAID_OK;
    hipLaunchKernelGGL(dropout_steps_kernel, grid1d(n * steps), dim3(256), 0, ST, x, y, n, steps, x_step, p, INVK(p), DS(seed));
    return check_launch("dropout_steps");
}
extern "C" int ekaid_act_bwd(const float* dy, const float* y, float* dz, long long n, int act, void* stream) {
    hipLaunchKernelGGL(act_bwd_kernel, grid1d(n), dim3(256), 0, ST, dy, y, dz, n, act);
    return check_launch("act_bwd");
}
extern "C" int ekaid_dropout(const float* x, float* y, long long n, float p, uint64_t seed, void* stream) {
    if (p < 0.f || p >= 1.f) { set_error("dropout: p=%f out of range", p); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(dropout_kernel, grid1d(n), dim3(256), 0, ST, x, y, n, p, INVK(p), DS(seed));
    return check_launch("dropout");
}
extern "C" int ekaid_axpby(const float* x, const float* y, float* out, long long n, float a, float b, void* stream) {
    hipLaunchKernelGGL(axpby_kernel, grid1d(n), dim3(256), 0, ST, x, y, out, n, a, b);
    return check_launch("axpby");
}
extern "C" int ekaid_self_mix3(const float* x, float* out, long long n, float a, float b, float c, void* stream) {
    hipLaunchKernelGGL(self_mix3_kernel, grid1d(n), dim3(256), 0, ST, x, out, n, a, b, c);
    return check_launch("self_mix3");
}
extern "C" int ekaid_rowmask_dropout(const float* x, long long rows, int cols, float* mask, float* y, float p, uint64_t seed, void* stream) {
    if (!(p >= 0.f && p < 1.f) || cols < 1) { set_error("rowmask_dropout: bad arguments"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(rowmask_dropout_kernel, grid1d(rows * 64), dim3(256), 0, ST, x, rows, cols, mask, y, p, INVK(p), DS(seed));
    return check_launch("rowmask_dropout");
}
extern "C" int ekaid_mul(const float* x, const float* y, float* out, long long n, void* stream) {
    hipLaunchKernelGGL(mul_kernel, grid1d(n), dim3(256), 0, ST, x, y, out, n);
    return check_launch("mul");
}
extern "C" int ekaid_rowsum(const float* x, long long rows, int cols, float* out, int as_nonzero_mask, void* stream) {
    if (cols <= 32) hipLaunchKernelGGL(rowsum_short_kernel, grid1d(rows), dim3(256), 0, ST, x, rows, cols, out, as_nonzero_mask);
    else hipLaunchKernelGGL(rowsum_kernel, grid1d(rows * 64), dim3(256), 0, ST, x, rows, cols, out, as_nonzero_mask);
    return check_launch("rowsum");
}
extern "C" int ekaid_expand_rows_fwd(const float* q, const float* mask, long long rows, int group, int qrows, int cols, float* out,
                                     float p, uint64_t seed, void* stream) {
    if (group < 1 || qrows < 1 || rows % group) { set_error("expand_rows: rows %% group != 0"); return EKAID_ERR_ARG; }
    if ((cols & 3) == 0 && rows > 0 && rows < (1ll << 31) && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(expand_rows_vec_kernel, dim3((unsigned)rows), dim3(256), 0, ST, q, mask, group, qrows, cols, out, p, INVK(p), DS(seed));
    else
        hipLaunchKernelGGL(expand_rows_kernel, grid1d(rows * cols), dim3(256), 0, ST, q, mask, rows, group, qrows, cols, out, p, INVK(p), DS(seed));
    return check_launch("expand_rows_fwd");
}
extern "C" int ekaid_expand_rows_bwd(const float* dout, const float* mask, long long rows, int group, int qrows, int cols, float* dq,
                                     float p, uint64_t seed, void* stream) {
    if (group < 1 || qrows < 1 || rows % group) { set_error("expand_rows: rows %% group != 0"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(expand_rows_bwd_kernel, grid1d((long long)qrows * cols), dim3(256), 0, ST, dout, mask, rows, group, qrows, cols, dq,
                       p, INVK(p), DS(seed));
    return check_launch("expand_rows_bwd");
}
extern "C" int ekaid_residual_relu2_fwd(const float* v, const float* o, float* vnew, long long n, float p, uint64_t seed, void* stream) {
    hipLaunchKernelGGL(residual_relu2_fwd_kernel, grid1d(n), dim3(256), 0, ST, v, o, vnew, n, p, INVK(p), DS(seed));
    return check_launch("residual_relu2_fwd");
}
extern "C" int ekaid_residual_relu2_bwd(const float* o, const float* dvnew, float* d_o, long long n, float p, uint64_t seed, void* stream) {
    hipLaunchKernelGGL(residual_relu2_bwd_kernel, grid1d(n), dim3(256), 0, ST, o, dvnew, d_o, n, p, INVK(p), DS(seed));
    return check_launch("residual_relu2_bwd");
}
extern "C" int ekaid_gate_fuse_fwd(const float* c2, const float* g2, const float* c1, const float* g1, float* out, long long n,
                                   long long n1, float p, uint64_t seed_c, uint64_t seed_g, void* stream) {
    if (n1 < 1 || n % n1) { set_error("gate_fuse: n %% n1 != 0"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(gate_fuse_fwd_kernel, grid1d(n), dim3(256), 0, ST, c2, g2, c1, g1, out, n, n1, p, INVK(p), DS(seed_c), DS(seed_g));
    return check_launch("gate_fuse_fwd");
}
extern "C" int ekaid_gate_fuse_bwd(const float* c2, const float* g2, const float* c1, const float* g1, const float* dout, float* dc2,
                                   float* dg2, float* dc1, float* dg1, long long n, long long n1, float p, uint64_t seed_c,
                                   uint64_t seed_g, void* stream) {
    if (n1 < 1 || n % n1) { set_error("gate_fuse: n %% n1 != 0"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(gate_fuse_bwd_kernel, grid1d(n1), dim3(256), 0, ST, c2, g2, c1, g1, dout, dc2, dg2, dc1, dg1, n, n1, p, INVK(p),
                       DS(seed_c), DS(seed_g));
    return check_launch("gate_fuse_bwd");
}
extern "C" int ekaid_weighted_pool_fwd(const float* w, const float* x, int G, int n, int C, float* out, void* stream) {
    hipLaunchKernelGGL(weighted_pool_fwd_kernel, grid1d((long long)G * C), dim3(256), 0, ST, w, x, G, n, C, out);
    return check_launch("weighted_pool_fwd");
}
extern "C" int ekaid_weighted_pool_bwd(const float* w, const float* x, const float* dout, int G, int n, int C, float* dw,
                                       float* dx, void* stream) {
    hipLaunchKernelGGL(weighted_pool_bwd_kernel, grid1d((long long)G * n * 64), dim3(256), 0, ST, w, x, dout, G, n, C, dw, dx);
    return check_launch("weighted_pool_bwd");
}
extern "C" int ekaid_qatt_weights_fwd(const float* atten, int B, int L, float* S_out, float* weight, void* stream) {
    hipLaunchKernelGGL(qatt_weights_fwd_kernel, dim3(L), dim3(64), 0, ST, atten, B, L, S_out, weight);
    return check_launch("qatt_weights_fwd");
}
extern "C" int ekaid_qatt_weights_bwd(const float* S, const float* dweight, int B, int L, float* datten, void* stream) {
    hipLaunchKernelGGL(qatt_weights_bwd_kernel, dim3(L), dim3(64), 0, ST, S, dweight, B, L, datten);
    return check_launch("qatt_weights_bwd");
}
extern "C" int ekaid_embedding_fwd(const int64_t* tok, long long ntok, const float* table, int E, float* out, int ld_out,
                                   int col_off, int relu, void* stream) {
    hipLaunchKernelGGL(embedding_fwd_kernel, grid1d(ntok * E), dim3(256), 0, ST, tok, ntok, table, E, out, ld_out, col_off, relu);
    return check_launch("embedding_fwd");
}
extern "C" int ekaid_embedding_bwd(const int64_t* tok, long long ntok, const float* table, int vocab, const float* dout, int E,
                                   int ld_out, int col_off, int relu, long long padding_idx, float* dtable, void* stream) {
    if (vocab <= 0 || ntok <= 0 || E <= 0) return EKAID_OK;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab, cdiv(E, 64)), dim3(64), 0, ST, tok, ntok, table, dout, E, ld_out, col_off,
                       relu, padding_idx, dtable);
    return check_launch("embedding_bwd");
}
extern "C" int ekaid_onehot(const int64_t* tok, long long ntok, int vocab, long long padding_idx, float* out, void* stream) {
    if (ntok <= 0 || vocab <= 0) return EKAID_OK;
    hipLaunchKernelGGL(onehot_kernel, grid1d(ntok * vocab), dim3(256), 0, ST, tok, ntok, vocab, padding_idx, out);
    return check_launch("onehot");
}
extern "C" int ekaid_masked_add(const float* raw, const float* table, int relu, float* dtable, long long n, void* stream) {
    if (n <= 0) return EKAID_OK;
    hipLaunchKernelGGL(masked_add_kernel, grid1d(n), dim3(256), 0, ST, raw, table, relu, dtable, n);
    return check_launch("masked_add");
}
extern "C" int ekaid_softmax_fwd(const float* x, float* y, long long rows, int cols, int log_mode, void* stream) {
    hipLaunchKernelGGL(softmax_fwd_kernel, grid1d(rows * 64), dim3(256), 0, ST, x, y, rows, cols, log_mode);
    return check_launch("softmax_fwd");
}
extern "C" int ekaid_softmax_bwd(const float* y, const float* dy, float* dx, long long rows, int cols, int log_mode, void* stream) {
    hipLaunchKernelGGL(softmax_bwd_kernel, grid1d(rows * 64), dim3(256), 0, ST, y, dy, dx, rows, cols, log_mode);
    return check_launch("softmax_bwd");
}
extern "C" int ekaid_mix3_fwd(const float* w, const float* a, const float* b, const float* c, int B, int C, float* out, void* stream) {
    hipLaunchKernelGGL(mix3_fwd_kernel, grid1d((long long)B * C), dim3(256), 0, ST, w, a, b, c, B, C, out);
    return check_launch("mix3_fwd");
}
extern "C" int ekaid_mix3_bwd(const float* w, const float* a, const float* b, const float* c, const float* dout, int B, int C,
                              float* dw, float* da, float* db, float* dc, int accumulate, void* stream) {
    hipLaunchKernelGGL(mix3_bwd_kernel, grid1d((long long)B * 64), dim3(256), 0, ST, w, a, b, c, dout, B, C, dw, da, db, dc, accumulate);
    return check_launch("mix3_bwd");
}
extern "C" int ekaid_nll_fwd(const float* logp, const int64_t* target, int ld_target, const float* mask, int ld_mask, int B, int T,
                             int V, const float* denom_dev, float* sums, float* loss_out, void* stream) {
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(1024), 0, ST, logp, target, ld_target, mask, ld_mask, B, T, V, denom_dev, sums, loss_out);
    return check_launch("nll_fwd");
}
extern "C" int ekaid_total_loss_fwd(const float* speaker_loss, const float* att_bef, const float* att_aft, long long n, float coef,
                                    float* out, void* stream) {
    hipLaunchKernelGGL(total_loss_kernel, dim3(1), dim3(256), 0, ST, speaker_loss, att_bef, att_aft, n, coef, out);
    return check_launch("total_loss_fwd");
}
extern "C" int ekaid_fill_scaled(const float* scalar_dev, float coef, float* out, long long n, void* stream) {
    hipLaunchKernelGGL(fill_scaled_kernel, grid1d(n), dim3(256), 0, ST, scalar_dev, coef, out, n);
    return check_launch("fill_scaled");
}
extern "C" int ekaid_nll_bwd(const int64_t* target, int ld_target, const float* mask, int ld_mask, const float* sums,
                             const float* gscale_dev, int B, int T, int V, float* dlogp, void* stream) {
    hipLaunchKernelGGL(nll_bwd_kernel, grid1d((long long)B * T * V), dim3(256), 0, ST, target, ld_target, mask, ld_mask, sums,
                       gscale_dev, B, T, V, dlogp);
    return check_launch("nll_bwd");
}
extern "C" int ekaid_layernorm_fwd(const float* x, const float* gamma, const float* beta, long long rows, int cols, float eps,
                                   float* y, float* save, void* stream) {
    hipLaunchKernelGGL(layernorm_fwd_kernel, grid1d(rows * 64), dim3(256), 0, ST, x, gamma, beta, rows, cols, eps, y, save);
    return check_launch("layernorm_fwd");
}
extern "C" int ekaid_layernorm_bwd(const float* x, const float* gamma, const float* save, const float* dy, long long rows, int cols,
                                   float* dx, float* dgamma, float* dbeta, void* stream) {
    hipLaunchKernelGGL(layernorm_bwd_kernel, grid1d(rows * 64), dim3(256), 0, ST, x, gamma, save, dy, rows, cols, dx, dgamma, dbeta);
    double* scr = reduction_scratch();
    if (!scr) { set_error("layernorm_bwd: scratch alloc failed"); return EKAID_ERR_LAUNCH; }
    float* pg = reinterpret_cast<float*>(scr + 2 * RED_SLOT);
    const int colblks = cdiv(cols, 256);
    int nparts = (int)(rows < 128 ? rows : 128);
    while (nparts > 1 && 2ll * nparts * cols > COLSUM_PART_FLOATS) nparts >>= 1;
    if (2ll * nparts * cols > COLSUM_PART_FLOATS) { set_error("layernorm_bwd: %d columns exceed the reduction scratch", cols); return EKAID_ERR_UNSUPPORTED; }
    const int rows_per_blk = cdiv(rows, nparts);
    nparts = cdiv(rows, rows_per_blk);
    float* pb = pg + (long long)nparts * cols;
    hipLaunchKernelGGL(layernorm_param_partial_kernel, dim3(colblks, nparts), dim3(256), 0, ST, x, save, dy, rows, cols, rows_per_blk, pg, pb);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, ST, (const float*)pg, nparts, cols, dgamma, 0);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, ST, (const float*)pb, nparts, cols, dbeta, 0);
    return check_launch("layernorm_bwd");
}
extern "C" int ekaid_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, int step, float grad_scale, void* stream) {
    if (step < 1) { set_error("adam_step: step must be >= 1"); return EKAID_ERR_ARG; }
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, grid1d(n), dim3(256), 0, ST, p, g, m, v, n, (float)(lr / bc1), beta1, beta2, eps,
                       (float)(1.0 / sqrt(bc2)), grad_scale);
    return check_launch("adam_step");
}
__global__ void seq_fault_flag_kernel(const unsigned* status, float* flag) { flag[0] = status[0] != 0u ? 1.f : 0.f; }
extern "C" int ekaid_seq_fault_flag(float* flag, void* stream) {
    if (!flag) { set_error("ekaid_seq_fault_flag: null pointer"); return EKAID_ERR_ARG; }
    const unsigned* status = seq_status_ptr();
    if (!status) { set_error("ekaid_seq_fault_flag: no status word (call ekaid_init first)"); return EKAID_ERR_LAUNCH; }
    hipLaunchKernelGGL(seq_fault_flag_kernel, dim3(1), dim3(1), 0, ST, status, flag);
    return check_launch("seq_fault_flag");
}
extern "C" int ekaid_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper_dev, int* step_dev,
                                   float* derived_dev, const float* fault_flag, void* stream) {
    if (!hyper_dev || !step_dev || !derived_dev) { set_error("adam_step_dev: null state pointer"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, ST, step_dev, hyper_dev, derived_dev, (const unsigned*)seq_status_ptr(), fault_flag);
    hipLaunchKernelGGL(adam_dev_kernel, grid1d(n), dim3(256), 0, ST, p, g, m, v, n, hyper_dev, (const float*)derived_dev);
    return check_launch("adam_step_dev");
}
extern "C" int ekaid_process_matrix(const void* adj, int label_type, int B, int S, int N, int L, float* out, void* stream) {
    if (N > S || N < 1 || L < 1) { set_error("process_matrix: bad dims N=%d S=%d L=%d", N, S, L); return EKAID_ERR_ARG; }
    if (label_type < 0 || label_type > 2) { set_error("process_matrix: label_type must be 0 (f64), 1 (i64) or 2 (i8)"); return EKAID_ERR_ARG; }
    const long long n = (long long)B * N * N * L;
    if (label_type == 2)
        hipLaunchKernelGGL(process_matrix_kernel<signed char>, grid1d(n), dim3(256), 0, ST, (const signed char*)adj, B, S, N, L, out);
    else if (label_type == 1)
        hipLaunchKernelGGL(process_matrix_kernel<long long>, grid1d(n), dim3(256), 0, ST, (const long long*)adj, B, S, N, L, out);
    else
        hipLaunchKernelGGL(process_matrix_kernel<double>, grid1d(n), dim3(256), 0, ST, (const double*)adj, B, S, N, L, out);
    return check_launch("process_matrix");
}

// ------------------------------------------------------------------ tiled transpose (weights -> [in][out] for the forward GEMM)
namespace ekaid {
__global__ __launch_bounds__(256) void transpose_kernel(const float* src, int R, int C, int lds_, float* dst, int ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(long long)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(long long)c * ldd + r] = tile[tx][i];
    }
}
}  // namespace ekaid
extern "C" int ekaid_transpose(const float* src, int R, int C, int ld_src, float* dst, int ld_dst, void* stream) {
    if (R <= 0 || C <= 0) return EKAID_OK;
    dim3 grid(ekaid::cdiv(C, 64), ekaid::cdiv(R, 64));
    hipLaunchKernelGGL(ekaid::transpose_kernel, grid, dim3(256), 0, ST, src, R, C, ld_src, dst, ld_dst);
    return ekaid::check_launch("transpose");
}

// ------------------------------------------------------------------ greedy decode bookkeeping (one step, whole batch)
// ref: dynamic_speaker_change_pos.py:316-352 with sample_max=1: forbid token 0 at t == 0, argmax, stop flags,
// it = it * unfinished, seq[:, t] = it, seq_logprobs[:, t] = max log-prob.  One wave per sample.
namespace ekaid {
__global__ __launch_bounds__(256) void greedy_step_kernel(const float* logp, int B, int V, int t, int* unfinished, int64_t* it,
                                                          int64_t* seq, float* seq_logp, int seq_len, int* n_unfinished) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    const float* p = logp + (long long)b * V;
    float best = -INFINITY;
    int arg = V;
    for (int v = lane; v < V; v += 64) {
        float x = p[v];
        if (t == 0 && v == 0) x = -INFINITY;
        if (x > best) { best = x; arg = v; }            // strict '>' keeps the lowest index within a lane
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }     // ties -> lowest index (torch.max)
    }
    if (lane == 0) {
        int un = (t == 0) ? (arg > 0) : (unfinished[b] && arg > 0);
        unfinished[b] = un;
        const int64_t tok = un ? arg : 0;
        it[b] = tok;
        seq[(long long)b * seq_len + t] = tok;
        seq_logp[(long long)b * seq_len + t] = best;
        if (un) atomicAdd(n_unfinished, 1);
    }
}
}  // namespace ekaid
extern "C" int ekaid_greedy_step(const float* logp, int B, int V, int t, int* unfinished, int64_t* it, int64_t* seq,
                                 float* seq_logp, int seq_len, int* n_unfinished, void* stream) {
    hipLaunchKernelGGL(ekaid::greedy_step_kernel, ekaid::grid1d((long long)B * 64), dim3(256), 0, ST, logp, B, V, t, unfinished, it,
                       seq, seq_logp, seq_len, n_unfinished);
    return ekaid::check_launch("greedy_step");
}

// ------------------------------------------------------------------ strided bias + activation + dropout, and its backward
namespace ekaid {
__global__ __launch_bounds__(256) void bias_act_drop_kernel(const float* x, int ldx, const float* bias, int rows, int cols, int act,
                                                            float p, float inv_keep, DropSeed seed, float* out) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        float v = x[(long long)r * ldx + c] + (bias ? bias[c] : 0.f);
        out[i] = act_apply(v, act) * dropout_scale(seed, (uint64_t)i, p, inv_keep);
    }
}
__global__ __launch_bounds__(256) void drop_act_bwd_kernel(const float* dy, const float* y, int ldy, int rows, int cols, int act, float p,
                                                           float inv_keep, DropSeed seed, float* out, int ldo) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        const float k = dropout_scale(seed, (uint64_t)i, p, inv_keep);
        out[(long long)r * ldo + c] = dy[i] * k * act_grad_from_out(y[(long long)r * ldy + c], act);
    }
}
}  // namespace ekaid
extern "C" int ekaid_bias_act_drop(const float* x, int ldx, const float* bias, int rows, int cols, int act, float p, uint64_t seed,
                                   float* out, void* stream) {
    hipLaunchKernelGGL(ekaid::bias_act_drop_kernel, ekaid::grid1d((long long)rows * cols), dim3(256), 0, ST, x, ldx, bias, rows, cols, act, p,
                       INVK(p), DS(seed), out);
    return ekaid::check_launch("bias_act_drop");
}
extern "C" int ekaid_drop_act_bwd(const float* dy, const float* y, int ldy, int rows, int cols, int act, float p, uint64_t seed,
                                  float* out, int ldo, void* stream) {
    hipLaunchKernelGGL(ekaid::drop_act_bwd_kernel, ekaid::grid1d((long long)rows * cols), dim3(256), 0, ST, dy, y, ldy, rows, cols, act, p,
                       INVK(p), DS(seed), out, ldo);
    return ekaid::check_launch("drop_act_bwd");
}

// ------------------------------------------------------------------ debug: count of non-finite values (EKAID_CHECK_FINITE=1)
namespace ekaid {
__global__ __launch_bounds__(256) void nonfinite_count_kernel(const float* __restrict__ x, long long n, int* count) {
    int c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned u = __float_as_uint(x[i]);
        c += ((u & 0x7F800000u) == 0x7F800000u) ? 1 : 0;         // Inf or NaN
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);      // integer atomics: order-independent
}
}  // namespace ekaid
extern "C" int ekaid_nonfinite_count(const float* x, long long n, int* count, void* stream) {
    if (!count) { ekaid::set_error("nonfinite_count: null counter"); return EKAID_ERR_ARG; }
    if (n <= 0 || !x) return EKAID_OK;
    hipLaunchKernelGGL(ekaid::nonfinite_count_kernel, ekaid::grid1d(n, 4), dim3(256), 0, ST, x, n, count);
    return ekaid::check_launch("nonfinite_count");
}
