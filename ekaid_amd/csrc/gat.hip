// Graph-attention core of GraphSelfAttentionLayer (ref: graph_att_layer.py:105-172) for the
// difference graph: per (sample, image) R-slice, N <= 64 region nodes attend over M <= 64 neighbours.
//
// gat_fused_fwd_kernel: one workgroup per R-slice, three phases, nothing but q/k/S in and P/outT out:
//   1. affinities  aff[n][h][m] = <q[n,h,:], k[m,h,:]> / sqrt(dh): q/k column chunks staged through LDS
//      (16-B coalesced HBM loads), 2x2 register-blocked dot products, accumulators live across chunks;
//   2. edge softmax: + log(max(geo,1e-6)) (implicit relation), adjacency mask with -9e15 (NOT -inf:
//      an isolated node gets a uniform row), + label bias AFTER masking, softmax over the <= 64
//      neighbours of a node held one-per-lane in a wave64 (shuffle max/sum reductions);
//   3. aggregation outT[n][h][:] = sum_m P[n][h][m] * S[m][:]: every thread owns VEC feature columns,
//      keeps the whole neighbour tile S[0..M)[cols] in VGPRs (read from HBM exactly once, 16 B/lane)
//      and streams the P rows from LDS as broadcast ds_read_b128.
// The kernel is HBM-bound: algorithmic bytes per slice = 3*N*A*4 (q,k,S) + N*H*A*4 (outT) + N*H*M*4 (P).
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <cstdlib>

namespace ekaid {

#define NEG_MASK (-9e15f)
#define GEO_EPS 1e-6f

__device__ __forceinline__ void edge_inputs(const float* cond, const float* lbias, long long rbase, int N, int n, int m,
                                            int dir, float& c, float& lb) {
    const long long idx = dir ? rbase + (long long)m * N + n : rbase + (long long)n * N + m;
    c = cond[idx];
    lb = lbias ? lbias[idx] : 0.f;
}

// softmax of one (n,h) row held one element per lane (lanes >= M inactive)
__device__ __forceinline__ float row_softmax(float x, bool active) {
    const float xm = active ? x : -INFINITY;
    const float mx = wave_max(xm);
    const float e = active ? expf(x - mx) : 0.f;
    const float s = wave_sum(e);
    return e / s;
}

// The next q/k column chunk is fetched into registers while the current one is multiplied out (without it every chunk paid a
// full, un-hidden HBM round trip between two barriers): -3 % at N = 36, -6.5 % at N = 52, -15 % at 64 slices
// (profiles/r03_gat_experiments.txt, which also lists what did NOT help: a per-head pipeline and LDS-staged edge terms).
template <int VEC, int MMAX>
__global__ __launch_bounds__(256) void gat_fused_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ S, const float* __restrict__ cond,
                                                            const float* __restrict__ lbias, const float* __restrict__ geo,
                                                            int N, int M, int A, int H, int dir, int DC,
                                                            float* __restrict__ P, float* __restrict__ outT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int MP = (M + 3) & ~3;
    const int LDC = DC + 4;
    float* aff_s = smem;                       // [N*H][MP]
    float* qs = aff_s + N * H * MP;            // [N][LDC]
    float* ks = qs + N * LDC;                  // [M][LDC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long r = blockIdx.x;
    // query-node split (gridDim.y parts per slice, used when there are fewer slices than CUs): this workgroup owns nodes
    // [n_lo, n_lo + NL) -- their affinities, softmax rows and output rows; keys and values are read whole by every part
    const int n_per = (N + gridDim.y - 1) / gridDim.y;
    const int n_lo = blockIdx.y * n_per;
    const int NL = min(N - n_lo, n_per);
    if (NL <= 0) return;
    const int dh = A / H;
    const float scale = 1.0f / sqrtf((float)dh);
    const float* qr = q + (r * N + n_lo) * A;
    const float* kr = k + r * M * A;           // keys exist for the first M (= nongt) nodes only
    const float* Sr = S + r * N * A;

    // ---------------- phase 1: scaled affinities into LDS
    const int n2 = (NL + 1) >> 1, m2 = (M + 1) >> 1, tiles = n2 * m2;
    float acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = 0.f;

    const int dc4 = DC >> 2;
    constexpr int QV = 4, KV = (MMAX * 16 + 255) / 256;       // float4 per thread per chunk (N <= 64, DC <= 64)
    f32x4 pq[QV], pk[KV];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < QV; ++j) {
            const int i = tid + j * 256;
            if (i < NL * dc4) pq[j] = *reinterpret_cast<const f32x4*>(&qr[(long long)(i / dc4) * A + c0 + 4 * (i % dc4)]);
        }
#pragma unroll
        for (int j = 0; j < KV; ++j) {
            const int i = tid + j * 256;
            if (i < M * dc4) pk[j] = *reinterpret_cast<const f32x4*>(&kr[(long long)(i / dc4) * A + c0 + 4 * (i % dc4)]);
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < A; c0 += DC) {
#pragma unroll
        for (int j = 0; j < QV; ++j) {
            const int i = tid + j * 256;
            if (i < NL * dc4) *reinterpret_cast<f32x4*>(&qs[(i / dc4) * LDC + 4 * (i % dc4)]) = pq[j];
        }
#pragma unroll
        for (int j = 0; j < KV; ++j) {
            const int i = tid + j * 256;
            if (i < M * dc4) *reinterpret_cast<f32x4*>(&ks[(i / dc4) * LDC + 4 * (i % dc4)]) = pk[j];
        }
        __syncthreads();
        if (c0 + DC < A) fetch(c0 + DC);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int t = tid + p * 256;
            if (t < tiles) {
                const int na = 2 * (t / m2), ma = 2 * (t % m2);
                const int nb = min(na + 1, NL - 1), mb = min(ma + 1, M - 1);
                const float* qa = qs + na * LDC; const float* qb = qs + nb * LDC;
                const float* ka = ks + ma * LDC; const float* kb = ks + mb * LDC;
                for (int d = 0; d < DC; d += 4) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(qa + d);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(qb + d);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(ka + d);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(kb + d);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[p][0] = fmaf(a0[e], b0[e], acc[p][0]);
                        acc[p][1] = fmaf(a0[e], b1[e], acc[p][1]);
                        acc[p][2] = fmaf(a1[e], b0[e], acc[p][2]);
                        acc[p][3] = fmaf(a1[e], b1[e], acc[p][3]);
                    }
                }
            }
        }
        if ((c0 + DC) % dh == 0) {             // a head is complete: park it in LDS, reset accumulators
            const int hd = c0 / dh;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int t = tid + p * 256;
                if (t < tiles) {
                    const int na = 2 * (t / m2), ma = 2 * (t % m2);
                    aff_s[(na * H + hd) * MP + ma] = acc[p][0] * scale;
                    if (ma + 1 < M) aff_s[(na * H + hd) * MP + ma + 1] = acc[p][1] * scale;
                    if (na + 1 < NL) {
                        aff_s[((na + 1) * H + hd) * MP + ma] = acc[p][2] * scale;
                        if (ma + 1 < M) aff_s[((na + 1) * H + hd) * MP + ma + 1] = acc[p][3] * scale;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[p][j] = 0.f;
            }
        }
        __syncthreads();
    }

    // ---------------- phase 2: edge softmax, one (n,h) row per wave pass, neighbour m on lane m
    const long long ebase = r * N * N;
    for (int row = wave; row < NL * H; row += 4) {
        const int n = n_lo + row / H, hd = row % H;
        const bool active = lane < M;
        float x = 0.f;
        if (active) {
            float c, lb;
            edge_inputs(cond, lbias, ebase, N, n, lane, dir, c, lb);
            x = aff_s[row * MP + lane];
            if (geo) x += logf(fmaxf(geo[((r * N + n) * M + lane) * H + hd], GEO_EPS));
            x = (c > 0.f ? x : NEG_MASK) + lb;
        }
        const float pv = row_softmax(x, active);
        if (active) P[((r * N + n) * H + hd) * M + lane] = pv;
        if (lane < MP) aff_s[row * MP + lane] = active ? pv : 0.f;
    }
    __syncthreads();

    // ---------------- phase 3: neighbour aggregation, thread-owned feature columns
    for (int cb = tid * VEC; cb < A; cb += 256 * VEC) {
        float s[MMAX][VEC];
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            if (m < M) {
                if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Sr[(long long)m * A + cb]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s[m][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s[m][e] = Sr[(long long)m * A + cb + e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[m][e] = 0.f;
            }
        }
        for (int row = 0; row < NL * H; ++row) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = 0.f;
            const float* prow = aff_s + row * MP;
#pragma unroll
            for (int m4 = 0; m4 < MMAX; m4 += 4) {
                if (m4 < MP) {
                    const f32x4 pw = *reinterpret_cast<const f32x4*>(prow + m4);   // same address in all lanes: broadcast
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o[e] = fmaf(pw[j], s[m4 + j][e], o[e]);
                }
            }
            float* op = outT + (((r * N + n_lo) * H + row) * (long long)A) + cb;
            if (VEC == 4) {
                f32x4 v = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(op) = v;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) op[e] = o[e];
            }
        }
    }
}

// ---------------- stand-alone edge softmax (used with the GEMM-composed affinities, and by every backward)
__global__ __launch_bounds__(256) void masked_softmax_fwd_kernel(const float* aff, const float* cond, const float* lbias,
                                                                 const float* geo, int R, int N, int M, int H, int dir, float* P) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long rows = (long long)R * N;
    for (long long rn = wave0; rn < rows; rn += nwaves) {
        const long long r = rn / N;
        const int n = (int)(rn % N);
        const bool active = lane < M;
        float c = 0.f, lb = 0.f;
        if (active) edge_inputs(cond, lbias, r * N * N, N, n, lane, dir, c, lb);
        for (int hd = 0; hd < H; ++hd) {
            float x = 0.f;
            if (active) {
                x = aff[(rn * H + hd) * M + lane];
                if (geo) x += logf(fmaxf(geo[(rn * M + lane) * H + hd], GEO_EPS));
                x = (c > 0.f ? x : NEG_MASK) + lb;
            }
            const float pv = row_softmax(x, active);
            if (active) P[(rn * H + hd) * M + lane] = pv;
        }
    }
}

__global__ __launch_bounds__(256) void masked_softmax_bwd_kernel(const float* P, const float* dP, const float* cond,
                                                                 const float* geo, int R, int N, int M, int H, int dir,
                                                                 float* daff, float* dlbias, float* dgeo) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long rows = (long long)R * N;
    for (long long rn = wave0; rn < rows; rn += nwaves) {
        const long long r = rn / N;
        const int n = (int)(rn % N);
        const bool active = lane < M;
        const long long eidx = dir ? r * N * N + (long long)lane * N + n : r * N * N + (long long)n * N + lane;
        const float c = active ? cond[eidx] : 0.f;
        float dlb = 0.f;
        for (int hd = 0; hd < H; ++hd) {
            const long long pidx = (rn * H + hd) * M + lane;
            const float p = active ? P[pidx] : 0.f;
            const float dp = active ? dP[pidx] : 0.f;
            const float dot = wave_sum(p * dp);
            const float dx = p * (dp - dot);
            dlb += dx;
            if (active) {
                daff[pidx] = c > 0.f ? dx : 0.f;
                if (dgeo) {
                    const long long gidx = (rn * M + lane) * H + hd;
                    const float gv = geo[gidx];
                    dgeo[gidx] = (c > 0.f && gv > GEO_EPS) ? dx / gv : 0.f;
                }
            }
        }
        if (active && dlbias) dlbias[eidx] = dlb;
    }
}
// ---------------- pair geometry: float64 math exactly as the reference, rounded to f32 once at the end
// ref: utils/mimic_utils.py:152-208.  out[r][i][j][c*(fd/4) + s*(fd/8) + t], c = component (dx,dy,dw,dh),
// s = 0 sin / 1 cos, t = frequency index; dim_mat[t] = 1000^(8t/fd) is the reference's float32 table, built on the host
// with the same torch expression (mimic_utils.py:195-197) so that no device powf rounding enters the argument.
__global__ __launch_bounds__(256) void geom_embed_kernel(const double* bb, const float* dim_mat, int R, int N, int M, int fd, float* out) {
    // one thread per (pair, component): the log-geometry value is computed once and fanned out to the fd/8 frequencies
    const int nf = fd / 8;
    const long long total = (long long)R * N * M * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long rest = i;
        const int comp = (int)(rest % 4); rest /= 4;
        const int j = (int)(rest % M); rest /= M;
        const int ii = (int)(rest % N);
        const long long r = rest / N;
        // M < N (nongt_dim below the node count): the reference keeps the first M ROWS of the N x N pair matrix
        // (mimic_utils.py:186) and then re-reads that [M][N] block as [N][M] (graph_att_layer.py:117-126): output position
        // (ii, j) holds the geometry of pair (p, q) with p * N + q = ii * M + j.  Reproduced as is.
        const int flat = ii * M + j;
        const int p = M < N ? flat / N : ii, qn = M < N ? flat % N : j;
        const double* bi = bb + (r * N + p) * 4;
        const double* bj = bb + (r * N + qn) * 4;
        const double wi = bi[2] - bi[0] + 1.0, hi = bi[3] - bi[1] + 1.0;
        const double wj = bj[2] - bj[0] + 1.0, hj = bj[3] - bj[1] + 1.0;
        double v;
        if (comp == 0) {
            v = fabs((0.5 * (bi[0] + bi[2]) - 0.5 * (bj[0] + bj[2])) / wi);
            v = log(v < 1e-3 ? 1e-3 : v);
        } else if (comp == 1) {
            v = fabs((0.5 * (bi[1] + bi[3]) - 0.5 * (bj[1] + bj[3])) / hi);
            v = log(v < 1e-3 ? 1e-3 : v);
        } else if (comp == 2) {
            v = log(wi / wj);
        } else {
            v = log(hi / hj);
        }
        float* o = out + (((r * N + ii) * M + j) * (long long)fd) + comp * (2 * nf);
        for (int t = 0; t < nf; ++t) {
            const double arg = (100.0 * v) / (double)dim_mat[t];
            double sn, cs;
            sincos(arg, &sn, &cs);
            o[t] = (float)sn;
            o[nf + t] = (float)cs;
        }
    }
}

}  // namespace ekaid

using namespace ekaid;
#define ST ((hipStream_t)stream)

template <int VEC, int MMAX>
static int launch_fused(const float* q, const float* k, const float* S, const float* cond, const float* lbias,
                        const float* geo, int R, int N, int M, int A, int H, int dir, int DC, size_t lds, float* P,
                        float* outT, hipStream_t st) {
    auto kern = gat_fused_fwd_kernel<VEC, MMAX>;
    if (lds > 64 * 1024) {
        if (func_attr_once(MMAX == 40 ? 48 : 49)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                set_error("gat_fused_fwd: cannot raise dynamic LDS limit to %zu bytes", lds);
                return EKAID_ERR_UNSUPPORTED;
            }
        }
    }
    // fewer slices than ~1.5 per CU (the strong-scaling batches): split the query nodes of a slice over up to 8 workgroups
    // (64 slices of 36 nodes: 4 parts 65 us, 8 parts 48 us; 128 slices of 52 nodes: 4 parts 134 us, 6-8 parts 168-185 us)
    static const char* force = getenv("EKAID_GAT_PARTS");       // tuning aid
    int parts = force ? atoi(force) : (R >= 384 ? 1 : (R >= 192 ? 2 : (R >= 96 ? 4 : 8)));
    parts = parts < 1 ? 1 : (parts > N ? N : parts);
    hipLaunchKernelGGL(kern, dim3(R, parts), dim3(256), lds, st, q, k, S, cond, lbias, geo, N, M, A, H, dir, DC, P, outT);
    return check_launch("gat_fused_fwd");
}

extern "C" int ekaid_gat_fused_fwd(const float* q, const float* k, const float* S, const float* cond, const float* lbias,
                                   const float* geo, int R, int N, int M, int A, int H, int dir, float* P, float* outT,
                                   void* stream) {
    if (R <= 0) return EKAID_OK;
    if (N < 1 || N > 64 || M < 1 || M > N || H < 1 || A % H || (A & 3) || ((A / H) & 3)) {
        set_error("gat_fused_fwd: unsupported shape N=%d M=%d A=%d H=%d", N, M, A, H);
        return EKAID_ERR_UNSUPPORTED;
    }
    const int dh = A / H;
    int DC = dh < 64 ? dh : 64;
    while (dh % DC) DC -= 4;
    const int MP = (M + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)N * H * MP + (size_t)(N + M) * (DC + 4));
    if (lds > 150 * 1024) { set_error("gat_fused_fwd: LDS need %zu too large", lds); return EKAID_ERR_UNSUPPORTED; }
    if (M <= 40) return launch_fused<4, 40>(q, k, S, cond, lbias, geo, R, N, M, A, H, dir, DC, lds, P, outT, ST);
    return launch_fused<2, 64>(q, k, S, cond, lbias, geo, R, N, M, A, H, dir, DC, lds, P, outT, ST);
}

static dim3 wave_grid(long long waves) {
    long long b = (waves + 3) / 4;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

extern "C" int ekaid_masked_softmax_fwd(const float* aff, const float* cond, const float* lbias, const float* geo, int R,
                                        int N, int M, int H, int dir, float* P, void* stream) {
    if (M > 64 || M > N) { set_error("masked_softmax: M=%d must be <= min(64, N)", M); return EKAID_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(masked_softmax_fwd_kernel, wave_grid((long long)R * N), dim3(256), 0, ST, aff, cond, lbias, geo, R, N, M, H, dir, P);
    return check_launch("masked_softmax_fwd");
}
extern "C" int ekaid_masked_softmax_bwd(const float* P, const float* dP, const float* cond, const float* geo, int R, int N,
                                        int M, int H, int dir, float* daff, float* dlbias, float* dgeo, void* stream) {
    if (M > 64 || M > N) { set_error("masked_softmax: M=%d must be <= min(64, N)", M); return EKAID_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, wave_grid((long long)R * N), dim3(256), 0, ST, P, dP, cond, geo, R, N, M, H, dir,
                       daff, dlbias, dgeo);
    return check_launch("masked_softmax_bwd");
}
extern "C" int ekaid_geom_embed(const double* bb, const float* dim_mat, int R, int N, int M, int feat_dim, float* out, void* stream) {
    if (feat_dim % 8) { set_error("geom_embed: feat_dim %% 8 != 0"); return EKAID_ERR_ARG; }
    const long long total = (long long)R * N * M * 4;
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(geom_embed_kernel, dim3((unsigned)b), dim3(256), 0, ST, bb, dim_mat, R, N, M, feat_dim, out);
    return check_launch("geom_embed");
}
