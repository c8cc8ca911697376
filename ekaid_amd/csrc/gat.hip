// Graph-attention core of GraphSelfAttentionLayer (ref: graph_att_layer.py:105-172) for the
// difference graph: per (sample, image) R-slice, N <= 64 region nodes attend over M <= 64 neighbours.
//
// ekaid_gat_fused_fwd = two launches, nothing but q/k/S in and P/outT out:
//   1. affinities  aff[r][n][h][m] = <q[n,h,:], k[m,h,:]> / sqrt(dh) of ALL (slice, head) pairs as ONE batched product on the
//      matrix pipe (R x H problems of N x M x dh through ekaid_gemm_f32, written into P in its final layout).  Round 3: the
//      LDS-staged f32 FMA version of this phase inside the kernel below took 5-20 % of the kernel's time more than the product.
//   gat_fused_fwd_kernel, one workgroup per R-slice (or per group of query nodes of a slice):
//   2. edge softmax: + log(max(geo,1e-6)) (implicit relation), adjacency mask with -9e15 (NOT -inf:
//      an isolated node gets a uniform row), + label bias AFTER masking, softmax over the <= 64
//      neighbours of a node held one-per-lane in a wave64 (shuffle max/sum reductions); each affinity is read from P and
//      overwritten with its probability by the same lane, the rows are kept in LDS (row stride MMAX, zero padded);
//   3. aggregation outT[n][h][:] = sum_m P[n][h][m] * S[m][:]: every thread owns VEC feature columns,
//      keeps the whole neighbour tile S[0..MMAX)[cols] in VGPRs (read from HBM exactly once, 16 B/lane, rows >= M zero)
//      and streams the P rows from LDS as broadcast ds_read_b128.  Everything in the row loop has compile-time bounds: the
//      reads of a row go out together and the FMAs follow as one straight-line block (with run-time bounds on the neighbour
//      count the compiler emitted one [ds_read, wait, 8 FMAs, branch] block per four neighbours: 2.8x the FMA issue time).
// Algorithmic bytes per slice = 3*N*A*4 (q,k,S) + N*H*A*4 (outT) + N*H*M*4 (P).
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <cstdlib>
#include <cstring>

namespace ekaid {

typedef float f32x2 __attribute__((ext_vector_type(2)));

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

template <int VEC, int MMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gat_fused_fwd_kernel(const float* __restrict__ S, const float* __restrict__ cond,
                                                            const float* __restrict__ lbias, const float* __restrict__ geo,
                                                            int N, int M, int A, int H, int dir,
                                                            float* __restrict__ P, float* __restrict__ outT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(MMAX % 4 == 0 && MMAX <= 64, "neighbour tile");
    float* aff_s = smem;                       // [NL*H][MMAX] probabilities, columns >= M zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long r = blockIdx.x;
    // query-node split (gridDim.y parts per slice, used when there are fewer slices than CUs): this workgroup owns nodes
    // [n_lo, n_lo + NL) -- their softmax rows and output rows; the values are read whole by every part
    const int n_per = (N + gridDim.y - 1) / gridDim.y;
    const int n_lo = blockIdx.y * n_per;
    const int NL = min(N - n_lo, n_per);
    if (NL <= 0) return;
    const float* Sr = S + r * N * A;

    // the neighbour tile of this thread's first column block does not depend on the softmax: requested first, it arrives under the
    // softmax phase (rows wider than 256 * VEC columns reload it per further block below)
    float s[MMAX][VEC];
    auto load_tile = [&](int cb) {
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            const float* sp = &Sr[(long long)min(m, M - 1) * A + cb];     // clamped row, zeroed below: no branch between the loads
            if (VEC == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[m][e] = m < M ? v[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[m][e] = m < M ? sp[e] : 0.f;
            }
        }
    };
    if (tid * VEC < A) load_tile(tid * VEC);

    // ---------------- edge softmax, one (n,h) row per wave pass, neighbour m on lane m.  Four rows per iteration: their inputs
    // (adjacency, label bias, affinity, geometry) are all requested before the first softmax is computed -- one exposed memory
    // round trip per four rows instead of one per row (the loads of a row cannot be hoisted above the P store of the previous one
    // by the compiler: P is read and written)
    const long long ebase = r * N * N;
    constexpr int RU = 4;
    for (int row0 = wave; row0 < NL * H; row0 += 4 * RU) {
        float xin[RU], cin[RU], lbin[RU], gin[RU];
        const bool active = lane < M;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = row0 + 4 * u;
            const bool ok = active && row < NL * H;
            const int rr = min(row, NL * H - 1);
            const int n = n_lo + rr / H, hd = rr % H;
            const long long pidx = ((r * N + n) * H + hd) * M + min(lane, M - 1);
            float c = 0.f, lb = 0.f;
            edge_inputs(cond, lbias, ebase, N, n, min(lane, M - 1), dir, c, lb);
            cin[u] = c; lbin[u] = lb;
            xin[u] = P[pidx];                                            // scaled affinity left there by the batched product
            gin[u] = geo ? geo[((r * N + n) * M + min(lane, M - 1)) * H + hd] : 1.f;
            (void)ok;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = row0 + 4 * u;
            if (row < NL * H) {                                          // wave-uniform
                const int n = n_lo + row / H, hd = row % H;
                float x = xin[u];
                if (geo) x += logf(fmaxf(gin[u], GEO_EPS));
                x = (cin[u] > 0.f ? x : NEG_MASK) + lbin[u];
                const float pv = row_softmax(x, active);
                if (active) P[((r * N + n) * H + hd) * M + lane] = pv;
                if (lane < MMAX) aff_s[row * MMAX + lane] = active ? pv : 0.f;
            }
        }
    }
    __syncthreads();

    // ---------------- neighbour aggregation, thread-owned feature columns
    for (int cb = tid * VEC; cb < A; cb += 256 * VEC) {
        if (cb != tid * VEC) load_tile(cb);
        float* op = outT + (((r * N + n_lo) * H) * (long long)A) + cb;
        for (int row = 0; row < NL * H; ++row, op += A) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = 0.f;
            const float* prow = aff_s + row * MMAX;
            f32x4 pw[MMAX / 4];
#pragma unroll
            for (int m4 = 0; m4 < MMAX / 4; ++m4) pw[m4] = *reinterpret_cast<const f32x4*>(prow + 4 * m4);   // same address in all lanes: broadcast
#pragma unroll
            for (int m4 = 0; m4 < MMAX / 4; ++m4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] = fmaf(pw[m4][j], s[4 * m4 + j][e], o[e]);
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

// ---------------- gradients of the affinity product:  dq[n][h,:] = sc * sum_m daff[n][h][m] k[m][h,:],
//                                                     dk[m][h,:] = sc * sum_n daff[n][h][m] q[n][h,:]
// As batched products these are R x H problems of 36 x 256 x 36: a 36-deep contraction on 64-deep tiles, 80 us each at 512 slices.
// Here one workgroup per slice keeps daff (and its transpose) in LDS, every thread owns four feature columns (so its head is
// fixed), holds the k / q column tile in registers and walks the output rows with compile-time inner bounds, exactly like the
// forward aggregation.  f32 FMAs; k, q read once, dq, dk written once.
template <int MMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gat_dqk_kernel(
    const float* __restrict__ daff, const float* __restrict__ q, const float* __restrict__ k, int N, int M, int A, int H, float sc,
    float* __restrict__ dq, float* __restrict__ dk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(MMAX % 4 == 0 && MMAX <= 64, "node tile");
    float* dA = smem;                          // [N*H][MMAX]   rows (n,h), columns m (>= M zero)
    float* dAT = smem + N * H * MMAX;          // [M*H][MMAX]   rows (m,h), columns n (>= N zero)
    const int tid = threadIdx.x;
    const long long r = blockIdx.x;
    const float* dr = daff + r * N * H * M;
    for (int i = tid; i < N * H * MMAX; i += 256) dA[i] = 0.f;
    for (int i = tid; i < M * H * MMAX; i += 256) dAT[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < N * H * M; i += 256) {
        const int m = i % M, nh = i / M, h = nh % H, n = nh / H;
        const float v = dr[i];
        dA[nh * MMAX + m] = v;
        dAT[(m * H + h) * MMAX + n] = v;
    }
    __syncthreads();
    const int dh = A / H;
    for (int cb = tid * 4; cb < A; cb += 1024) {
        const int h = cb / dh;
        float t[MMAX][4];
        // ---- dq: rows n, contraction over the neighbours m
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&k[(r * M + min(m, M - 1)) * A + cb]);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[m][e] = m < M ? v[e] : 0.f;
        }
        float* op = dq + r * N * A + cb;
        for (int n = 0; n < N; ++n, op += A) {
            const float* prow = dA + (n * H + h) * MMAX;
            f32x4 pw[MMAX / 4];
#pragma unroll
            for (int j = 0; j < MMAX / 4; ++j) pw[j] = *reinterpret_cast<const f32x4*>(prow + 4 * j);
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MMAX / 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf(pw[j][u], t[4 * j + u][e], o[e]);
            f32x4 v = {o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc};
            *reinterpret_cast<f32x4*>(op) = v;
        }
        // ---- dk: rows m, contraction over the query nodes n
#pragma unroll
        for (int n = 0; n < MMAX; ++n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&q[(r * N + min(n, N - 1)) * A + cb]);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[n][e] = n < N ? v[e] : 0.f;
        }
        op = dk + r * M * A + cb;
        for (int m = 0; m < M; ++m, op += A) {
            const float* prow = dAT + (m * H + h) * MMAX;
            f32x4 pw[MMAX / 4];
#pragma unroll
            for (int j = 0; j < MMAX / 4; ++j) pw[j] = *reinterpret_cast<const f32x4*>(prow + 4 * j);
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < MMAX / 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf(pw[j][u], t[4 * j + u][e], o[e]);
            f32x4 v = {o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc};
            *reinterpret_cast<f32x4*>(op) = v;
        }
    }
}

// ---------------- backward of the aggregation and of the edge softmax, one workgroup per slice (round 6):
//   dP[nh][m]  = sum_c d(outT)[nh][c] S[m][c]          (K = A)
//   dS[m][c]   = sum_nh P[nh][m] d(outT)[nh][c]        (K = N H)
//   daff = mask * P (dP - <P, dP>),  d(label bias) = sum over heads,  d(geometry) = daff / geo
// As batched products the first two were R problems of (N H) x M x A and M x A x (N H) with M = 36 rows or columns on 64-wide tiles
// (120 + 140 us at 512 slices), each reading d(outT) -- 302 MB of the backward's 0.77 GB -- on its own, followed by the softmax
// kernel.  Here d(outT) is read ONCE, in chunks of CW feature columns staged through LDS (CW * 4-byte row segments: 256 bytes at
// CW = 64; 128-byte segments measured the same: the bare load loop moves 4.2 TB/s either way, half as many barriers at 64); both
// products run on v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain per accumulator):
//   * dP accumulates over the chunks in registers: its (N H / 16) x (M / 16) tiles are dealt round-robin to the four waves;
//   * dS of a chunk is complete within the chunk: the waves split its K = N H range (the P fragments of a wave's k-steps do not
//     depend on the chunk: loaded once, kept in registers), their partial tiles meet in LDS (in the buffer the chunk was staged
//     in) and leave as CW * 4-byte row segments;
//   * the next chunk's loads are in flight while the current one is multiplied.
// After the last chunk dP goes to LDS and the softmax backward runs on it, a wave per query node (its H rows share the adjacency
// row and the label-bias sum), exactly as masked_softmax_bwd_kernel.  MT = neighbour tiles (16 columns each), TPW = dP tiles per wave.
template <int MT, int TPW, int CW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gat_bwd_core_kernel(const float* __restrict__ doutT, const float* __restrict__ S,
                                                           const float* __restrict__ P, const float* __restrict__ dPext,
                                                           const float* __restrict__ cond, const float* __restrict__ geo,
                                                           int N, int M, int A, int H, int dir,
                                                           float* __restrict__ dS, float* __restrict__ daff,
                                                           float* __restrict__ dlbias, float* __restrict__ dgeo, int csoff) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = CW + 2;                 // row stride of the staged chunk: 16 rows x 2 k-columns of an A-type fragment hit 32 banks
    constexpr int PLD = 16 * MT + 1;           // row stride of dP in LDS (odd: a thread per row walks it without bank conflicts)
    constexpr int RLD = CW + 1;
    constexpr int RTMAX = 4 * TPW / MT;        // row tiles the accumulators cover
    constexpr int V4 = CW / 4;                 // 16-byte vectors per staged row
    constexpr int DU = (RTMAX * 16 * V4 + 255) / 256;            // 16-byte loads per thread and chunk: d(outT) ...
    constexpr int SU = (MT * 16 * V4 + 255) / 256;               // ... and the values
    constexpr int KPW = (RTMAX * 4 + 3) / 4;   // k-steps (4 rows each) of the dS product per wave
    const int NH = N * H, NHp = (NH + 15) & ~15, RT = NHp >> 4;
    float* Dc = smem;                          // [NHp][LD]     d(outT) chunk (rows >= N H zero)
    float* Sc = Dc + NHp * LD;                 // [16 MT][LD]   value chunk (rows >= M zero)
    float* red = Dc;                           // [4][16 MT][RLD]  partial dS tiles of the four waves (after the chunk's products)
    float* dPs = Dc;                           // [NHp][PLD]    dP (after the last chunk)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const long long r = blockIdx.x;
    const float* Dr = doutT + r * NH * (long long)A;
    const float* Sr = S + r * N * (long long)A;
    const float* Pr = P + r * NH * (long long)M;

    f32x4 dreg[DU], sreg[SU];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int idx = tid + 256 * u, row = idx / V4, c4 = idx % V4;
            dreg[u] = row < NH ? *reinterpret_cast<const f32x4*>(Dr + (long long)row * A + c0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = tid + 256 * u, row = idx / V4, c4 = idx % V4;
            sreg[u] = row < M ? *reinterpret_cast<const f32x4*>(Sr + (long long)row * A + c0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int idx = tid + 256 * u, row = idx / V4, c4 = idx % V4;
            if (row < NHp) {
                float* d = Dc + row * LD + 4 * c4;                       // (row stride 8 mod 16 bytes: 8-byte aligned stores)
                *reinterpret_cast<f32x2*>(d) = f32x2{dreg[u][0], dreg[u][1]};
                *reinterpret_cast<f32x2*>(d + 2) = f32x2{dreg[u][2], dreg[u][3]};
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = tid + 256 * u, row = idx / V4, c4 = idx % V4;
            if (row < 16 * MT) {
                float* d = Sc + row * LD + 4 * c4;
                *reinterpret_cast<f32x2*>(d) = f32x2{sreg[u][0], sreg[u][1]};
                *reinterpret_cast<f32x2*>(d + 2) = f32x2{sreg[u][2], sreg[u][3]};
            }
        }
    };
    // chunk order: slice r starts at chunk (5 r) mod nch and wraps.  The rows of d(outT) are 16 KB apart and the slices a multiple of that:
    // walked in the same order by every workgroup, the chunk loads of the whole grid start on the same few memory channels; staggered
    // they spread (measured on the bare load loop of 512 slices: 164 -> 150 us -- the address hash already does most of it).  The
    // order a slice adds its chunks into dP depends on r only: deterministic.
    const int nch = A / CW;
    const int cstart = (int)((r * 5) % nch);
    fetch(cstart * CW);
    // P and the adjacency of the slice: contiguous copies in LDS for the whole kernel (the dS fragments below and the softmax backward
    // at the end read them), requested here so that they arrive under the first chunk
    float* cs = smem + csoff;                  // [N][N]        adjacency, as stored
    float* Pl = cs + N * N;                    // [NH][M + 1]   probabilities
    const int PM = M + 1;
    // flat index -> (row, column): n / d = umulhi(n, 2^32 / d + 1) for the small n, d here (an integer division is ~40 instructions)
    const unsigned mgM = M > 1 ? 0xFFFFFFFFu / (unsigned)M + 1u : 0u, mgH = H > 1 ? 0xFFFFFFFFu / (unsigned)H + 1u : 0u,
                   mgN = N > 1 ? 0xFFFFFFFFu / (unsigned)N + 1u : 0u;
    auto divM = [&](int i) { return M > 1 ? (int)__umulhi((unsigned)i, mgM) : i; };
    auto divH = [&](int i) { return H > 1 ? (int)__umulhi((unsigned)i, mgH) : i; };
    auto divN = [&](int i) { return N > 1 ? (int)__umulhi((unsigned)i, mgN) : i; };
    {
        constexpr int UB = 8;                  // loads in flight per thread (a loop of load -> store pays one round trip per element)
        for (int i0 = tid; i0 < NH * M; i0 += 256 * UB) {
            float v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = i0 + 256 * u < NH * M ? Pr[i0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + 256 * u, row = divM(i);
                if (i < NH * M) Pl[row * PM + (i - row * M)] = v[u];
            }
        }
        for (int i0 = tid; i0 < N * N; i0 += 256 * UB) {
            float v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = i0 + 256 * u < N * N ? cond[r * N * N + i0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (i0 + 256 * u < N * N) cs[i0 + 256 * u] = v[u];
        }
    }
    __syncthreads();
    // the P^T fragments of this wave's k-steps (k-step ks = wave + 4 i covers rows 4 ks .. 4 ks + 3 of P): A[i = m][k = row];
    // they do not depend on the chunk: registers
    float pa[KPW][MT];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int krow = 4 * (wave + 4 * i) + l4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * mt + l15;
            pa[i][mt] = (krow < NH && m < M) ? Pl[krow * PM + m] : 0.f;
        }
    }
    // this wave's dP tiles: t = wave + 4 j -> (row tile t / MT, neighbour tile t % MT); fragment base addresses in the staged chunk
    f32x4 accP[TPW];
    int aoff[TPW], boff[TPW];
    const int ntiles = RT * MT;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = min(wave + 4 * j, ntiles - 1);
        aoff[j] = (16 * (t / MT) + l15) * LD + l4;
        boff[j] = (16 * (t % MT) + l15) * LD + l4;
        accP[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int ci = 0; ci < nch; ++ci) {
        const int c = ci + cstart < nch ? ci + cstart : ci + cstart - nch;
        stage();
        __syncthreads();
        if (ci + 1 < nch) fetch((c + 1 < nch ? c + 1 : 0) * CW);
        // ---- dP += d(outT)_c . S_c^T
        // (no branch inside: a wave with fewer than TPW tiles repeats its last one into an accumulator nobody stores -- with a
        // test per tile every product sat in a block of its own behind the full latency of its two LDS reads)
#pragma unroll 2
        for (int ks = 0; ks < CW / 4; ++ks)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
                accP[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Dc[aoff[j] + 4 * ks], Sc[boff[j] + 4 * ks], accP[j], 0, 0, 0);
        // ---- dS_c = P^T . d(outT)_c over this wave's quarter of the (node, head) rows
        f32x4 accS[MT][CW / 16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < CW / 16; ++c2) accS[mt][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int krow = min(4 * (wave + 4 * i) + l4, NHp - 1);       // (a k-step past the last row: its P fragment is zero)
            float b[CW / 16];
#pragma unroll
            for (int c2 = 0; c2 < CW / 16; ++c2) b[c2] = Dc[krow * LD + 16 * c2 + l15];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c2 = 0; c2 < CW / 16; ++c2) accS[mt][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[i][mt], b[c2], accS[mt][c2], 0, 0, 0);
        }
        __syncthreads();                                            // every wave is done with the staged chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c2 = 0; c2 < CW / 16; ++c2)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[(wave * 16 * MT + 16 * mt + 4 * l4 + v) * RLD + 16 * c2 + l15] = accS[mt][c2][v];
        __syncthreads();
        for (int i = tid; i < M * CW; i += 256) {
            const int m = i / CW, cc = i - m * CW;
            const float* q0 = red + m * RLD + cc;
            const float sum = (q0[0] + q0[16 * MT * RLD]) + (q0[2 * 16 * MT * RLD] + q0[3 * 16 * MT * RLD]);
            dS[(r * N + m) * (long long)A + c * CW + cc] = sum;
        }
        __syncthreads();                                            // the partials are read: the buffer takes the next chunk
    }
    // ---- dP to LDS, then the softmax backward
#pragma unroll
    for (int j = 0; j < TPW; ++j)
        if (wave + 4 * j < ntiles) {
            const int t = wave + 4 * j, rt = t / MT, ct = t % MT;
#pragma unroll
            for (int v = 0; v < 4; ++v) dPs[(16 * rt + 4 * l4 + v) * PLD + 16 * ct + l15] = accP[j][v];
        }
    __syncthreads();
    // ---- softmax backward on the slice's N H rows.  (First form: a wave per query node, one row per pass with shuffle reductions and
    // 4-byte strided accesses -- 46 us of the kernel with 8 waves per CU to hide them.)  Now: P and the adjacency come in as contiguous
    // copies, a THREAD owns a row (dot product and dx in registers / LDS, no cross-lane step), and every output leaves in flat index
    // order from LDS.
    if (tid < NH) {
        float* dpr = dPs + tid * PLD;
        const float* pr = Pl + tid * PM;
        if (dPext) for (int m = 0; m < M; ++m) dpr[m] += dPext[(r * NH + tid) * M + m];
        float dot = 0.f;
#pragma unroll 4
        for (int m = 0; m < M; ++m) dot = fmaf(pr[m], dpr[m], dot);
#pragma unroll 4
        for (int m = 0; m < M; ++m) dpr[m] = pr[m] * (dpr[m] - dot);          // dx, unmasked (the label-bias gradient sums it as it is)
    }
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < NH * M; i += 256) {                                // daff[r][n][h][m]
        const int row = divM(i), m = i - row * M, n = divH(row);
        const float c = cs[dir ? m * N + n : n * N + m];
        daff[r * NH * M + i] = c > 0.f ? dPs[row * PLD + m] : 0.f;
    }
    if (dgeo)
#pragma unroll 4
        for (int i = tid; i < N * M * H; i += 256) {                         // dgeo[r][n][m][h]
            const int nm = divH(i), hd = i - nm * H, n = divM(nm), m = nm - n * M;
            const float c = cs[dir ? m * N + n : n * N + m];
            const float gv = geo[r * N * M * H + i];
            dgeo[r * N * M * H + i] = (c > 0.f && gv > GEO_EPS) ? dPs[(n * H + hd) * PLD + m] / gv : 0.f;
        }
    if (dlbias)
        for (int i = tid; i < N * M; i += 256) {                             // dlbias[r][n][m] (dir 0) / [r][m][n] (dir 1): flat order of the output
            const int a = dir ? divN(i) : divM(i), b = i - a * (dir ? N : M);
            const int n = dir ? b : a, m = dir ? a : b;
            float sum = 0.f;
            for (int hd = 0; hd < H; ++hd) sum += dPs[(n * H + hd) * PLD + m];
            dlbias[r * N * N + (dir ? m * N + n : n * N + m)] = sum;
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
__global__ __launch_bounds__(256) void geom_embed_kernel(const double* bb, const float* dim_mat, int R, int N, int M, int fd, float* out,
                                                         float p, float inv_keep, DropSeed seed) {
    // one thread per (pair, component): the log-geometry value is computed once and fanned out to the fd/8 frequencies
    extern __shared__ __attribute__((aligned(16))) float tile[];        // [256 items][2 nf]
    const int nf = fd / 8;
    const long long total = (long long)R * N * M * 4;
    for (long long i0 = (long long)blockIdx.x * 256; i0 < total; i0 += (long long)gridDim.x * 256) {
        const long long i = i0 + threadIdx.x;
        if (i < total) {
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
        // item i owns out[i * 2nf .. (i + 1) * 2nf) (fd = 4 * 2nf): the workgroup's 256 items form ONE contiguous block of the output;
        // staged through LDS so that it leaves as full 16-byte-per-lane rows instead of 2nf scalar stores with a 8nf-byte lane stride
        float* o = tile + threadIdx.x * (2 * nf);
        for (int t = 0; t < nf; ++t) {
            // argument in float64 as the reference forms it; reduced to [-pi, pi] in float64 (two-term 2 pi), then ONE float32
            // sincos: |error| <= 2e-7 (rounding of the reduced argument) + 1 ulp, against 2e-6 allowed
            const double arg = (100.0 * v) / (double)dim_mat[t];
            const double kk = rint(arg * 0.15915494309189535);
            double red = fma(-kk, 6.283185307179586, arg);
            red = fma(-kk, 2.4492935982947064e-16, red);
            float sn, cs;
            sincosf((float)red, &sn, &cs);
            o[t] = sn;
            o[nf + t] = cs;
        }
        }
        __syncthreads();
        {
            const long long first = i0 * (2 * nf);                          // first output element of this workgroup's block
            const long long cnt = min((long long)256, total - i0) * (2 * nf);
            // p > 0: the input dropout of the layer that reads the embedding (graph_att_layer.py:121, fc.py:36-37), mask index = flat index
            if ((nf & 1) == 0) {
                for (long long e = 4 * threadIdx.x; e < cnt; e += 1024) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(tile + e);
                    if (p > 0.f) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] *= dropout_scale(seed, (uint64_t)(first + e + c), p, inv_keep);
                    }
                    *reinterpret_cast<f32x4*>(out + first + e) = v;
                }
            } else {
                for (long long e = threadIdx.x; e < cnt; e += 256) out[first + e] = tile[e] * dropout_scale(seed, (uint64_t)(first + e), p, inv_keep);
            }
        }
        __syncthreads();
    }
}

}  // namespace ekaid

using namespace ekaid;
#define ST ((hipStream_t)stream)

template <int VEC, int MMAX>
static int launch_fused(const float* S, const float* cond, const float* lbias, const float* geo, int R, int N, int M, int A, int H,
                        int dir, float* P, float* outT, hipStream_t st) {
    auto kern = gat_fused_fwd_kernel<VEC, MMAX>;
    // fewer slices than ~1.5 per CU (the strong-scaling batches): split the query nodes of a slice over up to 8 workgroups
    static const char* force = getenv("EKAID_GAT_PARTS");       // tuning aid
    int parts = force ? atoi(force) : (R >= 384 ? 1 : (R >= 192 ? 2 : (R >= 96 ? 4 : 8)));
    parts = parts < 1 ? 1 : (parts > N ? N : parts);
    const int n_per = (N + parts - 1) / parts;
    const size_t lds = sizeof(float) * (size_t)n_per * H * MMAX;
    if (lds > 64 * 1024) {
        if (func_attr_once((const void*)kern)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                set_error("gat_fused_fwd: cannot raise dynamic LDS limit to %zu bytes", lds);
                return EKAID_ERR_UNSUPPORTED;
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3(R, parts), dim3(256), lds, st, S, cond, lbias, geo, N, M, A, H, dir, P, outT);
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
    if ((size_t)N * H * 64 * sizeof(float) > 150 * 1024) { set_error("gat_fused_fwd: too many (node, head) rows for the LDS"); return EKAID_ERR_UNSUPPORTED; }
    // 1. scaled affinities of every (slice, head) pair into P[r][n][h][m]: one two-level batched product
    const int dh = A / H;
    ekaid_gemm_t g;
    memset(&g, 0, sizeof(g));
    g.A = q; g.B = k; g.C = P;
    g.M = N; g.N = M; g.K = dh;
    g.lda = A; g.ldb = A; g.ldc = H * M;
    g.transA = 0; g.transB = 1;
    g.batch1 = R; g.batch2 = H;
    g.strideA1 = (long long)N * A; g.strideA2 = dh;
    g.strideB1 = (long long)M * A; g.strideB2 = dh;
    g.strideC1 = (long long)N * H * M; g.strideC2 = M;
    g.alpha = 1.0f / sqrtf((float)dh);
    g.rdiv = 1;
    const int rc = ekaid_gemm_f32(&g, nullptr, 0, stream);
    if (rc) return rc;
    // 2.+3. softmax and aggregation
    // the aggregation runs at the f32 vector rate (160 v_fmac per output row and thread at 40 neighbour slots: 119-129 us for 512 slices of
    // 36 nodes = 24 GFLOP executed), so every padded neighbour slot is paid in full: tiles of exactly 36 (the metric's region count)
    // and 52 (the real files') next to the general ones
    if (M <= 36) return launch_fused<4, 36>(S, cond, lbias, geo, R, N, M, A, H, dir, P, outT, ST);
    if (M <= 40) return launch_fused<4, 40>(S, cond, lbias, geo, R, N, M, A, H, dir, P, outT, ST);
    if (M <= 52) return launch_fused<2, 52>(S, cond, lbias, geo, R, N, M, A, H, dir, P, outT, ST);
    if (M <= 56) return launch_fused<2, 56>(S, cond, lbias, geo, R, N, M, A, H, dir, P, outT, ST);
    return launch_fused<2, 64>(S, cond, lbias, geo, R, N, M, A, H, dir, P, outT, ST);
}

template <int MMAX>
static int launch_dqk(const float* daff, const float* q, const float* k, int R, int N, int M, int A, int H, float sc, float* dq, float* dk,
                      hipStream_t st) {
    auto kern = gat_dqk_kernel<MMAX>;
    const size_t lds = sizeof(float) * (size_t)(N + M) * H * MMAX;
    if (lds > 64 * 1024) { set_error("gat_dqk: LDS need %zu too large", lds); return EKAID_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(kern, dim3(R), dim3(256), lds, st, daff, q, k, N, M, A, H, sc, dq, dk);
    return check_launch("gat_dqk");
}

extern "C" int ekaid_gat_dqk(const float* daff, const float* q, const float* k, int R, int N, int M, int A, int H, float scale,
                             float* dq, float* dk, void* stream) {
    if (R <= 0) return EKAID_OK;
    if (!daff || !q || !k || !dq || !dk) { set_error("gat_dqk: null pointer"); return EKAID_ERR_ARG; }
    if (N < 1 || N > 64 || M < 1 || M > N || H < 1 || A % H || (A & 3) || ((A / H) & 3) ||
        (size_t)(N + M) * H * 64 * sizeof(float) > 150 * 1024) {
        set_error("gat_dqk: unsupported shape N=%d M=%d A=%d H=%d", N, M, A, H);
        return EKAID_ERR_UNSUPPORTED;
    }
    // up to 40 nodes only: at 52 nodes the register tile (56 x 4) and 93 KB of LDS leave one workgroup per CU and the two batched
    // products are faster (619 vs 641 us for the whole backward of 512 slices); at 36 nodes 447 -> 364 us
    if (N <= 36) return launch_dqk<36>(daff, q, k, R, N, M, A, H, scale, dq, dk, ST);
    if (N <= 40) return launch_dqk<40>(daff, q, k, R, N, M, A, H, scale, dq, dk, ST);
    set_error("gat_dqk: more than 40 nodes (N=%d): compose from batched products", N);
    return EKAID_ERR_UNSUPPORTED;
}

template <int MT, int TPW, int CW>
static int launch_bwd_core(const float* doutT, const float* S, const float* P, const float* dPext, const float* cond, const float* geo, int R,
                           int N, int M, int A, int H, int dir, float* dS, float* daff, float* dlbias, float* dgeo, hipStream_t st) {
    auto kern = gat_bwd_core_kernel<MT, TPW, CW>;
    const int NHp = (N * H + 15) & ~15;
    size_t area = (size_t)NHp * (CW + 2) + (size_t)16 * MT * (CW + 2);                 // the staged chunk ...
    if ((size_t)4 * 16 * MT * (CW + 1) > area) area = (size_t)4 * 16 * MT * (CW + 1);  // ... the four waves' partial dS tiles ...
    if ((size_t)NHp * (16 * MT + 1) > area) area = (size_t)NHp * (16 * MT + 1);        // ... dP after the last chunk: one buffer
    const size_t lds = sizeof(float) * (area + (size_t)N * N + (size_t)N * H * (M + 1));      // + the slice's adjacency and probabilities
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) { set_error("gat_bwd_core: LDS need %zu too large", lds); return EKAID_ERR_UNSUPPORTED; }
        if (func_attr_once((const void*)kern)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                set_error("gat_bwd_core: cannot raise dynamic LDS limit to %zu bytes", lds);
                return EKAID_ERR_UNSUPPORTED;
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3(R), dim3(256), lds, st, doutT, S, P, dPext, cond, geo, N, M, A, H, dir, dS, daff, dlbias, dgeo, (int)area);
    count_launch(EKAID_COUNT_GAT_BWD_CORE);
    return check_launch("gat_bwd_core");
}

extern "C" int ekaid_gat_bwd_core(const float* doutT, const float* S, const float* P, const float* dP_ext, const float* cond, const float* geo,
                                  int R, int N, int M, int A, int H, int dir, float* dS, float* daff, float* dlbias, float* dgeo,
                                  void* stream) {
    if (R <= 0) return EKAID_OK;
    if (!doutT || !S || !P || !cond || !dS || !daff) { set_error("gat_bwd_core: null pointer"); return EKAID_ERR_ARG; }
    if (N < 1 || N > 64 || M < 1 || M > N || H < 1 || (A & 31) || (geo == nullptr) != (dgeo == nullptr)) {
        set_error("gat_bwd_core: unsupported shape N=%d M=%d A=%d H=%d", N, M, A, H);
        return EKAID_ERR_UNSUPPORTED;
    }
    const int RT = (N * H + 15) >> 4;
    if (M <= 48 && RT <= 9) {
        if (!(A & 63)) return launch_bwd_core<3, 7, 64>(doutT, S, P, dP_ext, cond, geo, R, N, M, A, H, dir, dS, daff, dlbias, dgeo, ST);
        return launch_bwd_core<3, 7, 32>(doutT, S, P, dP_ext, cond, geo, R, N, M, A, H, dir, dS, daff, dlbias, dgeo, ST);
    }
    // (a <4, 13, 32> instantiation for 52 nodes x 4 heads was measured and removed: 255 registers with spills, one workgroup per CU,
    // 916 vs 631 us for the backward of 512 slices next to the composition)
    set_error("gat_bwd_core: %d (node, head) rows x %d neighbours outside the register tile (compose from batched products)", N * H, M);
    return EKAID_ERR_UNSUPPORTED;
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
    return ekaid_geom_embed_dropout(bb, dim_mat, R, N, M, feat_dim, out, 0.f, 0, stream);
}
extern "C" int ekaid_geom_embed_dropout(const double* bb, const float* dim_mat, int R, int N, int M, int feat_dim, float* out, float p,
                                        uint64_t seed, void* stream) {
    if (feat_dim % 8) { set_error("geom_embed: feat_dim %% 8 != 0"); return EKAID_ERR_ARG; }
    if (!(p >= 0.f && p < 1.f)) { set_error("geom_embed: dropout probability outside [0, 1)"); return EKAID_ERR_ARG; }
    const long long total = (long long)R * N * M * 4;
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    if (feat_dim > 256) { set_error("geom_embed: feat_dim > 256"); return EKAID_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(geom_embed_kernel, dim3((unsigned)b), dim3(256), 256 * (feat_dim / 4) * sizeof(float), ST, bb, dim_mat, R, N, M, feat_dim, out,
                       p, p > 0.f ? 1.f / (1.f - p) : 1.f, DropSeed{seed, dropout_epoch_ptr()});
    return check_launch("geom_embed");
}
