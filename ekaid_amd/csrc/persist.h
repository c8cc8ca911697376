// Device helpers shared by the persistent sequence kernels (csrc/persist.hip: decoder forward, csrc/persist_bwd.hip: decoder backward):
// device-scope seams, the fragment-major hand-off layout, the 16 x 16 MFMA job engine.  See the header comment of persist.hip.
#pragma once
#include "common.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <cstring>

namespace ekaid {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSEAM = 8, SHARDS = 8, LINE = 32;                   // seams (5 per step, the prologue, 2 of greedy decoding); counter shards (one per XCD); words per 128-B line
constexpr int SYNC_WORDS = NSEAM * SHARDS * LINE + LINE;          // + one line for the status word
constexpr unsigned SPIN_LIMIT = 400000u;                          // polls (~1 us each) before a seam is declared dead
constexpr int PMAX_T = 96;                                        // steps per launch (seeds travel by value in the kernel arguments)
constexpr int MAXJ = 8;                                           // jobs of one phase per workgroup
constexpr int AUX_SC1 = 16;                                       // buffer cache-policy bit: sc1 (agent-scope write-through / L1 bypass)

// The workgroups of a persistent kernel wait for each other at the seams, so ALL of them must be resident at once.  Checked when a
// launch is PLANNED (not assumed): the runtime's occupancy of this kernel with this much dynamic LDS, times the CUs of the device, must
// cover the grid, and no CU mask may be in force (the CU-count attribute still reports the whole chip under a mask).  `false` sends
// the caller to the launch-per-product chain (csrc/sequence.hip).  What this cannot see -- another process on the same GPU, a resident
// collective kernel stalled on a late peer -- is caught at run time: bounded spins, the sticky status word, and the Adam kernel that
// refuses the update of a faulted step (pointwise.hip: adam_prep_kernel).
inline bool grid_fits(const void* kern, int G, int cus, size_t lds_bytes) {
    static const bool masked = getenv("HSA_CU_MASK") != nullptr || getenv("ROC_GLOBAL_CU_MASK") != nullptr;
    if (masked) return false;
    if (func_attr_once(kern)) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    // occupancy per (kernel, dynamic LDS, device): the LDS need varies with batch and sequence length, so the table is sized for every
    // persistent kernel x a few shapes x 8 devices; lookups are lock-free, inserts re-check under the mutex (no duplicate slots from
    // racing threads); a full table just asks the runtime again
    struct Ent { const void* k; size_t lds; int dev; int per_cu; };
    constexpr int CAP = 256;
    static Ent cache[CAP];
    static std::atomic<int> n_cache{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto find = [&](int n) -> int {
        for (int i = 0; i < n && i < CAP; ++i)
            if (cache[i].k == kern && cache[i].lds == lds_bytes && cache[i].dev == dev) return cache[i].per_cu;
        return -1;
    };
    int per_cu = find(n_cache.load(std::memory_order_acquire));
    if (per_cu < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        const int m = n_cache.load(std::memory_order_relaxed);
        if (find(m) < 0 && m < CAP) { cache[m] = Ent{kern, lds_bytes, dev, per_cu}; n_cache.store(m + 1, std::memory_order_release); }
    }
    return (long long)per_cu * cus >= G;
}

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }      // HW_REG_XCC_ID
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes), 0x00020000);
}
template <int AUX>
__device__ __forceinline__ f32x4 bload4(const __amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
__device__ __forceinline__ void bstore4_sc1(const __amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, AUX_SC1);
}
__device__ __forceinline__ void store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- seams ------------------------------------------------------------------------------------------------------------
// arrive: every wave has drained its (write-through) stores, the workgroup has met, one lane counts the workgroup in
__device__ __forceinline__ void seam_arrive(unsigned* seam, unsigned shard) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(seam + shard * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: wave 0 polls the eight shards (relaxed sc1 loads, one word per lane) until their sum reaches `target`; the barrier
// releases the other waves.  Returns false when the launch has to be abandoned (status word set).
__device__ __forceinline__ bool seam_wait(const unsigned* seam, unsigned target, unsigned* status, int* ok_lds, unsigned code) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool good = true;
        // two polls in flight: the sum examined in an iteration was requested one iteration earlier, so a poll leaves every half round trip
        const unsigned* src = seam + (lane < SHARDS ? lane : 0) * LINE;
        unsigned pend = lane < SHARDS ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        for (unsigned spins = 0;;) {
            const unsigned next = lane < SHARDS ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            unsigned v = pend;
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v = __shfl(v, 0, 64);
            if (v >= target) break;
            pend = next;
            ++spins;
            if ((spins & 127u) == 0u) {
                const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st != 0u || spins > SPIN_LIMIT) {
                    if (st == 0u && lane == 0) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = false;
                    break;
                }
            }
        }
        if (lane == 0) *ok_lds = good ? 1 : 0;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// ---- fragment-major hand-off layout ---------------------------------------------------------------------------------------
// X (rows x K) of row block rb: chunk (rb, kb) = the 16 rows x 16 k of k-block kb, 1 KB, element (row, k) at float offset
// ((k % 16) / 4 * 16 + row) * 4 + k % 4 -- lane l = kq * 16 + row of a wave reads its MFMA A-operand piece at byte l * 16.
__device__ __forceinline__ unsigned fm_chunk(int rb, int kb, int kblocks) { return (unsigned)(rb * kblocks + kb) << 10; }

// 4 waves' partial tiles -> sums.  red: [tile][wave][320 floats]; thread tid gets element (row = tid >> 4, col = tid & 15) of every
// tile.  One barrier for all NT tiles of a job; the caller alternates between two buffers from job to job.
template <int NT>
__device__ __forceinline__ void tiles_reduce(const f32x4 (&part)[NT], float (&v)[NT], float* red, int wave, int lane, int tid) {
    float* mine = red + wave * 320 + (lane >> 4) * 80 + (lane & 15);        // C layout: col = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int m = 0; m < NT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[m * 1280 + r * 16] = part[m][r];
    }
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const float* q = red + (row >> 2) * 80 + (row & 3) * 16 + col;
#pragma unroll
    for (int m = 0; m < NT; ++m) v[m] = (q[m * 1280] + q[m * 1280 + 320]) + (q[m * 1280 + 640] + q[m * 1280 + 960]);
}

struct Ctx {
    int tid, lane, wave, wg, G, nrb, B, row, col;
    float* red;
    int par;
    unsigned* status;
    int* ok_lds;
    unsigned long long* trace;       // NULL, or [16] stamps of the current step for this workgroup (thread 0 writes)
};
__device__ __forceinline__ void stamp(const Ctx& c, int k) {
    if (c.trace && c.tid == 0) c.trace[k] = __builtin_amdgcn_s_memrealtime();
}

// One MFMA phase of a step for this workgroup: jobs (tile group tg = wg % tgroups: MT column tiles sharing the activations; row blocks
// rb = wg / tgroups + n G / tgroups).
//   * the weight fragments of the MT tiles -- all NB blocks of this wave's K quarter, wload(tile m, block i) -- are requested BEFORE the seam is polled and stay
//     in registers for all jobs of the phase; so are the values pre(rb, n) every job's epilogue needs;
//   * activations stream from the fragment-major buffer through a ring of RX blocks that runs ACROSS jobs: as soon as block i of job n
//     has been multiplied, block i + RX is requested -- of the same job, or of job n + 1 -- so the reduction, the epilogue and the stores
//     of a job overlap the loads of the next;
//   * epi(v[MT], pre value, rb, n) after the reduction, with the tile elements of thread (row = tid >> 4, col = tid & 15).
template <int NB, int MT, class PreT, class WLoad, class Pre, class Epi>
__device__ __forceinline__ bool run_mt(Ctx& c, int tgroups, WLoad wload, const __amdgpu_buffer_rsrc_t rx, int kblocks,
                                       const unsigned* seam, unsigned target, unsigned code, int stamp_id, Pre pre, Epi epi) {
    constexpr int RX = MT >= 2 ? 8 : (NB < 16 ? NB : 16), NACC = MT >= 4 ? 1 : 4 / MT;
    static_assert(NB % RX == 0, "ring phase");
    const int njobs = c.wg < tgroups * c.nrb ? (tgroups * c.nrb - c.wg + c.G - 1) / c.G : 0;
    const int rb0 = c.wg / tgroups, drb = c.G / tgroups;
    f32x4 w[MT][NB], x[RX];
    PreT ev_next;
    if (njobs) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int i = 0; i < NB; ++i) w[m][i] = wload(m, i);
        }
        ev_next = pre(rb0, 0);
    }
    if (seam && !seam_wait(seam, target, c.status, c.ok_lds, code)) return false;
    if (stamp_id >= 0) stamp(c, stamp_id);
    if (!njobs) return true;
    const unsigned lo = (unsigned)c.lane * 16u;
    unsigned ao = fm_chunk(rb0, c.wave * NB, kblocks) + lo;
#pragma unroll
    for (int i = 0; i < RX; ++i) x[i] = bload4<AUX_SC1>(rx, ao + 1024u * i);
#pragma unroll 1
    for (int n = 0; n < njobs; ++n) {
        const int rb = rb0 + n * drb;
        const bool more = n + 1 < njobs;
        const unsigned ao2 = fm_chunk(rb + drb, c.wave * NB, kblocks) + lo;
        const PreT ev = ev_next;
        if (more) ev_next = pre(rb + drb, n + 1);        // the next job's epilogue operands travel under this job's matrix work
        f32x4 acc[MT][NACC] = {};
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = i % RX;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][e % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[sl][e], w[m][i][e], acc[m][e % NACC], 0, 0, 0);
            }
            const int ii = i + RX;
            if (ii < NB) x[sl] = bload4<AUX_SC1>(rx, ao + 1024u * ii);
            else if (more) x[sl] = bload4<AUX_SC1>(rx, ao2 + 1024u * (ii - NB));
        }
        ao = ao2;
        f32x4 part[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            part[m] = acc[m][0];
#pragma unroll
            for (int q = 1; q < NACC; ++q) part[m] += acc[m][q];
        }
        float v[MT];
        tiles_reduce<MT>(part, v, c.red + c.par * 5120, c.wave, c.lane, c.tid);
        c.par ^= 1;
        epi(v, ev, rb, n);
    }
    return true;
}

// gather the values of 4 consecutive lanes (cols 4q .. 4q+3 of a tile row) into the lane with col % 4 == 0
__device__ __forceinline__ f32x4 gather4(float v) {
    f32x4 r;
    r[0] = v; r[1] = __shfl_down(v, 1, 64); r[2] = __shfl_down(v, 2, 64); r[3] = __shfl_down(v, 3, 64);
    return r;
}

inline bool persist_trace_enabled() {
    static const bool on = [] { const char* e = getenv("EKAID_PERSIST_TRACE"); return e && e[0] == '1'; }();
    return on;
}
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline ekaid_gemm_t dense_desc(const float* A, int lda, const float* Bm, int ldb, int transB, float* C, int ldc, int M, int N, int K) {
    ekaid_gemm_t g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = Bm; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.transA = 0; g.transB = transB; g.batch1 = g.batch2 = 1; g.alpha = 1.f; g.rdiv = 1;
    return g;
}

}  // namespace
}  // namespace ekaid
