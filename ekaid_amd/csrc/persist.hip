// Persistent sequence kernels: the whole time loop of the teacher-forced answer decoder in ONE launch.
//
// ref: dynamic_speaker_change_pos.py:94-131 (DynamicCore.forward) inside :196-222 (the loop of DynamicSpeaker._forward).
//
// The step-wise path (csrc/sequence.hip) issues 8 dependent launches per decode step: four split-K slab products and four
// kernels that sum the slabs.  At 32-256 batch rows every one of them is latency-bound (7-14 us for < 1 us of arithmetic), and
// the chain is what the real-data loop, greedy decoding and the 8-GPU strong-scaling point spend their time on.  Here one
// workgroup per CU stays resident for all T steps:
//   * every product of a step is cut into jobs of 16 batch rows x 16 output columns with the FULL contraction (no split-K, no
//     slabs: each output element is produced by one workgroup in a fixed order -> deterministic by construction); the four waves
//     of the workgroup split K, exact f32 arithmetic on v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain per wave), operands loaded
//     global -> registers directly in fragment layout (16 bytes per lane, 64 contiguous bytes per row and instruction), the four
//     partial tiles summed through LDS;
//   * a workgroup owns the same (column tile, row block) pairs in every step, so LSTM cell states, and the h_lang-side partial
//     products a later phase of the step needs, never leave its LDS; weights are re-read from L2 each step, prefetched into
//     registers BEFORE the wait they hide behind;
//   * the five all-to-all dependencies of a step (h_att/vpos -> row-local module mix and POS chain -> gate-MLP layer 1 -> layer 2 ->
//     language LSTM -> next step) are device-scope seams instead of kernel boundaries: handed-off activations are stored write-through
//     (sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane adds to the seam's counter (sharded per
//     XCD, each shard on a line of its own); consumers poll the eight shards with sc1 loads from one wave, barrier, and then read the
//     handed-off bytes ONLY with sc1 loads (guide: cdna_hip_programming.md Guideline 16, form R1 / MI355X_MICROARCH.md "Valid forms",
//     first table row).  Counters count up over the steps of one launch (target = arrivals per step x steps done) and are zeroed by a
//     memset node in front of every launch; every activation buffer is indexed by the step, so no address is written twice in a
//     launch.  Spins are bounded: a seam that does not open within ~0.5 s sets the status word and every workgroup leaves.
// Work that only needs h_lang_{t-1} but is consumed late in the step (the h_lang blocks of the gate MLP and of the language LSTM)
// runs in the shadow between a seam's arrive and its wait.
// The saved-tensor contract is that of ekaid_speaker_seq_fwd (include/ekaid_hip.h): either forward pairs with either backward.
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <cstdlib>
#include <cstring>

namespace ekaid {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NSEAM = 5, SHARDS = 8, LINE = 32;                   // seams per step; counter shards (one per XCD); words per 128-B line
constexpr int SYNC_WORDS = NSEAM * SHARDS * LINE + LINE;          // + one line for the status word
constexpr unsigned SPIN_LIMIT = 400000u;                          // polls (~1 us each) before a seam is declared dead
constexpr int PMAX_T = 96;                                        // steps per launch (seeds travel by value in the kernel arguments)
#ifndef EKAID_PERSIST_ACQ
#define EKAID_PERSIST_ACQ 1
#endif
constexpr int AUX_SC1 = 16;
constexpr int AUX_RD = EKAID_PERSIST_ACQ ? 0 : 16;              // loads of handed-off bytes: plain behind an agent-scope acquire, or sc1                                       // buffer cache-policy bit: sc1 (agent-scope write-through / L1 bypass)

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
__device__ __forceinline__ float load_sc1(const float* p) {
    if (EKAID_PERSIST_ACQ) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
        for (unsigned spins = 0;;) {
            unsigned v = lane < SHARDS ? __hip_atomic_load(seam + lane * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v = __shfl(v, 0, 64);
            if (v >= target) break;
            ++spins;
            if ((spins & 127u) == 0u) {
                const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st != 0u || spins > SPIN_LIMIT) {
                    if (st == 0u && lane == 0) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = good ? 1 : 0;
        if (EKAID_PERSIST_ACQ) {        // ONE acquire after the match: drops this CU's stale L1 lines (and non-local L2 lines); the wait holds the barrier for it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    return *ok_lds != 0;
}

// ---- 16 x 16 x K tile jobs ----------------------------------------------------------------------------------------------
// Lane l of a wave holds, for the MFMA's A operand, 4 consecutive k of batch row (l & 15) and, for its B operand, the same 4 k of
// weight row (= output column) (l & 15); the four 16-lane groups take consecutive 4-k pieces, so one load instruction reads 64
// contiguous bytes of each of 16 rows.  Block i of a wave is 16 k further on; wave w of the workgroup owns k in [w K/4, (w+1) K/4).
template <int NB, int AUX>
__device__ __forceinline__ void load_frag(f32x4 (&v)[NB], const __amdgpu_buffer_rsrc_t r, unsigned off) {
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = bload4<AUX>(r, off + 64u * i);
}
// sum of the four waves' partial tiles; thread tid gets element (row = tid >> 4, col = tid & 15).  `red`: 4 x 320 floats; the
// caller alternates between two such buffers from job to job (one barrier per job)
__device__ __forceinline__ float tile_reduce(const f32x4 (&acc)[4], float* red, int wave, int lane, int tid) {
    const f32x4 c = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    float* mine = red + wave * 320 + (lane >> 4) * 80 + (lane & 15);        // C layout: col = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[r * 16] = c[r];
    __syncthreads();
    const int row = tid >> 4, col = tid & 15;
    const float* q = red + (row >> 2) * 80 + (row & 3) * 16 + col;
    return (q[0] + q[320]) + (q[640] + q[960]);
}

struct PFwd {
    int T, B, G, nrb;
    float p_lm, p_pos, inv_lm, inv_pos;
    const unsigned long long* ep;
    const float *Wcat, *whh_a, *bhh_a, *w_fc, *b_fc, *b_p1, *w_wp, *b_wp, *w_p2, *b_p2, *w_g1, *b_g1, *w_g2, *b_g2, *wih_l, *bhh_l;
    const float *bef, *diff, *aft, *GXA, *GXL;
    float *HA, *CA, *HL, *CL, *SA, *SL, *G1D, *MW, *VP, *DP, *PW, *PA, *AT, *GT, *GA;
    unsigned* sync;
    unsigned long long* trace;        // NULL or [G][T][16] stamps (s_memrealtime, 100 MHz)
    int We, jobs_max, rows_max;
    uint64_t seeds[3 * PMAX_T];
};

// LDS carve (floats); every offset a multiple of 4 floats
template <int R, int A, int P>
struct Lds {
    static constexpr int flag = 0;                              // 4
    static constexpr int red = 4;                               // 2 x 1280
    static constexpr int hrow = red + 2 * 1280;                 // R   (h_att row of phase B)
    static constexpr int vrow = hrow + R;                       // R   (vpos row)
    static constexpr int zpart = vrow + R;                      // 8   (module-weight partials: 2 waves x 3)
    static constexpr int pz = zpart + 8;                        // 128 (POS logit partials)
    static constexpr int sz = pz + 128;                         // 64  (POS weights)
    static constexpr int wfc = sz + 64;                         // 3 R
    static constexpr int bsm = wfc + 3 * R;                     // b_fc (4) | b_wp (P) | b_p2 (R)
    static constexpr int fixed = bsm + 4 + P + R;
    // then per job slot: g1p 256 | lhp 256 | catt 64 | clang 64; then per row slot: 3 A
    static constexpr int per_job = 640;
    static constexpr int per_row = 3 * A;
};

struct Ctx {
    int tid, lane, wave, wg, G, nrb, B, l15, row, col;
    float* red;
    int par;
    unsigned* status;
    int* ok_lds;
    unsigned long long* trace;       // NULL, or [16] stamps of the current step for this workgroup (thread 0 writes)
};
__device__ __forceinline__ void stamp(const Ctx& c, int k) {
    if (c.trace && c.tid == 0) c.trace[k] = __builtin_amdgcn_s_memrealtime();
}

// x-ring depth of a contraction of NB blocks per wave: at most 16 blocks (~1 us of matrix work) ahead, and a divisor of NB so that
// the ring keeps its phase from one job to the next
template <int NB> struct XRing { static constexpr int v = NB <= 16 ? NB : (NB % 16 == 0 ? 16 : (NB % 12 == 0 ? 12 : 8)); };

// One phase of a step for this workgroup: the jobs (column tile tau = wg % tiles, row blocks rb = wg / tiles + n G / tiles) it owns.
//   * the weight fragments of the tile -- ALL NB blocks of this wave's K quarter, NB0 from the first operand pair, NB1 from the second
//     -- are requested BEFORE the seam is polled and stay in registers for all jobs of the phase;
//   * activations stream through a ring of RX blocks that runs ACROSS jobs: as soon as block i of job n has been multiplied, block
//     i + RX is requested -- of the same job, or of job n + 1 -- so the reduction, the epilogue and the stores of a job overlap the
//     loads of the next;
//   * pre(rb, slot) is called before a job's matrix work (issue the loads the epilogue needs), epi(v, pre's value, rb, slot) after the
//     reduction with the tile element of thread (row = tid >> 4, col = tid & 15); slot = index of the job within the workgroup.
template <int NB0, int NB1, bool WRES, class Pre, class Epi>
__device__ __forceinline__ bool run_phase(Ctx& c, int tiles, const __amdgpu_buffer_rsrc_t rw0, unsigned woff0, const __amdgpu_buffer_rsrc_t rw1,
                                          unsigned woff1, const __amdgpu_buffer_rsrc_t ra0, const __amdgpu_buffer_rsrc_t ra1, unsigned lda_bytes,
                                          unsigned kw, const unsigned* seam, unsigned target, unsigned code, int stamp_id, Pre pre, Epi epi) {
    // WRES: all NB weight blocks resident in registers for the phase; otherwise (long contractions under the 256-register budget of
    // two workgroups per CU) the weights stream through a ring like the activations, one pass per job (from L2)
    constexpr int NB = NB0 + NB1, RX = XRing<NB>::v, NW = WRES ? NB : RX;
    static_assert(NB % RX == 0, "ring phase");
    const int njobs = c.wg < tiles * c.nrb ? (tiles * c.nrb - c.wg + c.G - 1) / c.G : 0;
    const int rb0 = c.wg / tiles, drb = c.G / tiles;
    f32x4 w[NW], x[RX];
    auto wload = [&](int i) { return i < NB0 ? bload4<0>(rw0, woff0 + kw + 64u * i) : bload4<0>(rw1, woff1 + kw + 64u * (i - NB0)); };
    if (njobs) {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = wload(i);
    }
    if (seam && !seam_wait(seam, target, c.status, c.ok_lds, code)) return false;
    if (stamp_id >= 0) stamp(c, stamp_id);
    if (!njobs) return true;
#ifdef EKAID_PERSIST_FAKE
    // EXPERIMENT (wrong results, same byte count): every load instruction of a wave reads 1 KB contiguous
#define XOFF(rbx) ((unsigned)min((rbx) * 16, c.B - 16) * lda_bytes + (unsigned)(c.wave * (NB * 1024) + c.lane * 16))
#define XSTEP 1024u
#else
#define XOFF(rbx) ((unsigned)min((rbx) * 16 + c.l15, c.B - 1) * lda_bytes + kw)
#define XSTEP 64u
#endif
    unsigned ao = XOFF(rb0);
#pragma unroll
    for (int i = 0; i < RX; ++i) x[i] = i < NB0 ? bload4<AUX_RD>(ra0, ao + XSTEP * i) : bload4<AUX_RD>(ra1, ao + XSTEP * (i - NB0));
    for (int n = 0; n < njobs; ++n) {
        const int rb = rb0 + n * drb;
        const bool more = n + 1 < njobs;
        const unsigned ao2 = XOFF(rb + drb);
        const auto ev = pre(rb, n);
        f32x4 acc[4] = {};
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sl = i % RX, wl = WRES ? i : sl;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[sl][e], w[wl][e], acc[e], 0, 0, 0);
            const int ii = i + RX;
            if (ii < NB) {
                x[sl] = ii < NB0 ? bload4<AUX_RD>(ra0, ao + XSTEP * ii) : bload4<AUX_RD>(ra1, ao + XSTEP * (ii - NB0));
                if (!WRES) w[wl] = wload(ii);
            } else if (more) {
                const int i2 = ii - NB;
                x[sl] = i2 < NB0 ? bload4<AUX_RD>(ra0, ao2 + XSTEP * i2) : bload4<AUX_RD>(ra1, ao2 + XSTEP * (i2 - NB0));
                if (!WRES) w[wl] = wload(i2);
            }
        }
        ao = ao2;
        const float v = tile_reduce(acc, c.red + c.par * 1280, c.wave, c.lane, c.tid);
        c.par ^= 1;
        epi(v, ev, rb, n);
    }
    return true;
}

template <int R, int A, int P>
__global__ __launch_bounds__(256, 1) void spk_seq_fwd_persistent_kernel(const PFwd a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<R, A, P>;
    constexpr int G1 = 2 * R + A, NW = 9 * R + G1;
    constexpr int NBR = R / 64, NBA = A / 64, NBPA = (R + A) / 64, NBG = G1 / 64;      // 16-k blocks per wave of a K-long contraction
    constexpr int TILES_U = R / 4;                               // gate tiles: 4 hidden units x 4 gates  (128)
    constexpr int TILES_P1 = R / 16, TILES_G1 = G1 / 16, TILES_G2 = A / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int B = a.B, G = a.G, nrb = a.nrb;
    const int l15 = lane & 15, kq4 = (lane >> 4) * 4;
    const int row = tid >> 4, col = tid & 15;                    // element of a reduced tile
    const unsigned shard = xcc_id();
    unsigned* const status = a.sync + NSEAM * SHARDS * LINE;
    float* const jobs_lds = lds + L::fixed;
    float* const rows_lds = jobs_lds + a.jobs_max * L::per_job;
    Ctx c = {tid, lane, wave, wg, G, nrb, B, l15, row, col, lds + L::red, 0, status, reinterpret_cast<int*>(lds + L::flag), nullptr};

    // column tiles of this workgroup (G is a multiple of every tile count: the tile never changes) and, per thread, the tile
    // element's output column, gate and hidden unit; the biases of those columns stay in registers for the whole sequence
    const int tU = wg % TILES_U, tP1 = wg % TILES_P1, tG1 = wg % TILES_G1, tG2 = wg % TILES_G2;
    const int gate = col >> 2, unit = 4 * tU + (col & 3);
    const int gate_row_l = (l15 >> 2) * R + 4 * tU + (l15 & 3);  // weight row of lane l15 in a gate tile (4 units x 4 gates)
    const float bA = a.bhh_a[gate * R + unit], bE = a.bhh_l[gate * R + unit], bP1 = a.b_p1[16 * tP1 + col], bG1 = a.b_g1[16 * tG1 + col],
                bG2 = a.b_g2[16 * tG2 + col];

    // ---- one-time preloads: row-local weights (phase B), feature rows and cell states of the pairs this workgroup owns
    for (int i = tid; i < 3 * R; i += 256) lds[L::wfc + i] = a.w_fc[i];
    if (tid < 3) lds[L::bsm + tid] = a.b_fc[tid];
    if (tid < P) lds[L::bsm + 4 + tid] = a.b_wp[tid];
    for (int i = tid; i < R; i += 256) lds[L::bsm + 4 + P + i] = a.b_p2[i];
    {
        int rs = 0;
        for (int r = wg; r < B; r += G, ++rs) {
            float* f = rows_lds + rs * L::per_row;
            for (int i = tid; i < A; i += 256) {
                f[i] = a.bef[(long long)r * A + i];
                f[A + i] = a.diff[(long long)r * A + i];
                f[2 * A + i] = a.aft[(long long)r * A + i];
            }
        }
        int slot = 0;
        for (int j = wg; j < TILES_U * nrb; j += G, ++slot) {
            const int rb = j / TILES_U;
            if (tid < 64) {
                const int grow = min(rb * 16 + (tid >> 2), B - 1), u = 4 * tU + (tid & 3);
                jobs_lds[slot * L::per_job + 512 + tid] = a.CA[(long long)grow * R + u];
                jobs_lds[slot * L::per_job + 576 + tid] = a.CL[(long long)grow * R + u];
            }
        }
    }
    __syncthreads();

    const long long BR = (long long)B * R;
    const __amdgpu_buffer_rsrc_t rsWcat = make_rsrc(a.Wcat, (long long)NW * R * 4), rsWhh = make_rsrc(a.whh_a, 4ll * R * R * 4),
                                 rsWg1 = make_rsrc(a.w_g1, (long long)G1 * G1 * 4), rsWg2 = make_rsrc(a.w_g2, (long long)A * G1 * 4),
                                 rsWil = make_rsrc(a.wih_l, 4ll * R * (a.We + A) * 4);
    const unsigned kwR = (unsigned)(wave * (R / 4) + kq4) * 4u;          // byte offset of this lane's first k in a K = R contraction
    const unsigned kwA = (unsigned)(wave * (A / 4) + kq4) * 4u;
    const unsigned kwPA = (unsigned)(wave * ((R + A) / 4) + kq4) * 4u;
    const unsigned kwG = (unsigned)(wave * (G1 / 4) + kq4) * 4u;
    const int ldw_l = a.We + A;
    auto nopre = [](int, int) { return 0.f; };

    for (int t = 0; t < a.T; ++t) {
        const float* HLp = a.HL + (long long)t * BR;             // h_lang_{t-1}
        const float* HAp = a.HA + (long long)t * BR;             // h_att_{t-1}
        float* HAn = a.HA + (long long)(t + 1) * BR;
        float* HLn = a.HL + (long long)(t + 1) * BR;
        const __amdgpu_buffer_rsrc_t rsHL = make_rsrc(HLp, BR * 4), rsHA = make_rsrc(HAp, BR * 4);
        const DropSeed sv = {a.seeds[3 * t], a.ep}, sd = {a.seeds[3 * t + 1], a.ep}, sg = {a.seeds[3 * t + 2], a.ep};
        const unsigned tgt = (unsigned)G * (unsigned)(t + 1);
        c.trace = a.trace ? a.trace + ((long long)wg * a.T + t) * 16 : nullptr;
        stamp(c, 0);

        // =================== phase A (critical part): attention-LSTM gates + cell, POS layer 1 ===================
        // att gates = [h_lang | h_att] . [W_ih_att[:, E:] | W_hh_att]^T: one contraction over two operand pairs
        if (!run_phase<NBR, NBR, true>(c, TILES_U, rsWcat, (unsigned)(gate_row_l * R) * 4u, rsWhh, (unsigned)(gate_row_l * R) * 4u, rsHL, rsHA, (unsigned)R * 4u,
                                 kwR, t > 0 ? a.sync + 4 * SHARDS * LINE : nullptr, (unsigned)G * (unsigned)t, 0x100u + 4, 1,
                                 [&](int rb, int) { return a.GXA[((long long)t * B + min(rb * 16 + row, B - 1)) * 4 * R + gate * R + unit]; },
                                 [&](float v, float gx, int rb, int slot) {
                                     const int grow = rb * 16 + row;
                                     const float pre = v + gx + bA;
                                     const float act = gate == 2 ? tanhf(pre) : sigm(pre);
                                     if (grow < B) a.SA[((long long)t * B + grow) * 4 * R + gate * R + unit] = act;
                                     const float fg = __shfl_down(act, 4, 64), gg = __shfl_down(act, 8, 64), og = __shfl_down(act, 12, 64);
                                     if (gate == 0) {
                                         float* cs = jobs_lds + slot * L::per_job + 512 + row * 4 + (col & 3);
                                         const float cn = fg * cs[0] + act * gg;
                                         const float hn = og * tanhf(cn);
                                         cs[0] = cn;
                                         if (grow < B) {
                                             a.CA[(long long)(t + 1) * BR + (long long)grow * R + unit] = cn;
                                             store_sc1(HAn + (long long)grow * R + unit, hn);
                                         }
                                     }
                                 }))
            return;
        stamp(c, 2);
        // POS layer 1: vpos = dropout(relu(h_lang W_pos1^T + b))
        run_phase<NBR, 0, true>(c, TILES_P1, rsWcat, (unsigned)((4 * R + 16 * tP1 + l15) * R) * 4u, rsWcat, 0u, rsHL, rsHL, (unsigned)R * 4u, kwR, nullptr, 0u, 0u,
                          -1, nopre, [&](float v, float, int rb, int) {
                              const int k = 16 * tP1 + col, grow = rb * 16 + row;
                              v += bP1;
                              v = (v > 0.f ? v : 0.f) * dropout_scale(sv, (uint64_t)grow * R + k, a.p_lm, a.inv_lm);
                              if (grow < B) store_sc1(a.VP + ((long long)t * B + grow) * R + k, v);
                          });
        stamp(c, 3);
        seam_arrive(a.sync + 0 * SHARDS * LINE, shard);
        stamp(c, 4);
        // ---- shadow of seam 0: the h_lang block of gate-MLP layer 1, kept in LDS for phase C
        run_phase<NBR, 0, true>(c, TILES_G1, rsWcat, (unsigned)((5 * R + 16 * tG1 + l15) * R) * 4u, rsWcat, 0u, rsHL, rsHL, (unsigned)R * 4u, kwR, nullptr, 0u, 0u,
                          -1, nopre, [&](float v, float, int, int slot) { jobs_lds[slot * L::per_job + tid] = v; });
        stamp(c, 5);
        if (!seam_wait(a.sync + 0 * SHARDS * LINE, tgt, status, c.ok_lds, 0x100u + 0)) return;
        stamp(c, 6);

        // =================== phase B: row-local module mix and POS chain ===================
        {
            float* PAt = a.PA + (long long)t * B * (R + A);
            float* ATt = a.AT + (long long)t * B * A;
            const __amdgpu_buffer_rsrc_t rsPA = make_rsrc(PAt, (long long)B * (R + A) * 4), rsAT = make_rsrc(ATt, (long long)B * A * 4);
            const __amdgpu_buffer_rsrc_t rh = make_rsrc(HAn, BR * 4), rv = make_rsrc(a.VP + (long long)t * BR, BR * 4);
            int rs = 0;
            for (int r = wg; r < B; r += G, ++rs) {
                if (rs) __syncthreads();
                if (tid < 128) *reinterpret_cast<f32x4*>(lds + L::hrow + 4 * tid) = bload4<AUX_RD>(rh, (unsigned)(r * R + 4 * tid) * 4u);
                else *reinterpret_cast<f32x4*>(lds + L::vrow + 4 * (tid - 128)) = bload4<AUX_RD>(rv, (unsigned)(r * R + 4 * (tid - 128)) * 4u);
                __syncthreads();
                if (tid < 128) {                                 // module-weight logits: 3 dot products over R
                    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
                    for (int k = tid; k < R; k += 128) {
                        const float x = lds[L::hrow + k];
                        z0 += x * lds[L::wfc + k]; z1 += x * lds[L::wfc + R + k]; z2 += x * lds[L::wfc + 2 * R + k];
                    }
                    z0 = wave_sum(z0); z1 = wave_sum(z1); z2 = wave_sum(z2);
                    if (lane == 0) { lds[L::zpart + wave * 4] = z0; lds[L::zpart + wave * 4 + 1] = z1; lds[L::zpart + wave * 4 + 2] = z2; }
                } else {                                         // POS logits: thread (j = q >> 3, c = q & 7) sums the k with (k / 4) % 8 == c
                    const int q = tid - 128, j = q >> 3, cq = q & 7;
                    const f32x4* wr = reinterpret_cast<const f32x4*>(a.w_wp + j * R) + cq;
                    float s = 0.f;
#pragma unroll 4
                    for (int i = 0; i < R / 32; ++i) {
                        const f32x4 wv = wr[8 * i], xv = *reinterpret_cast<const f32x4*>(lds + L::vrow + 4 * (cq + 8 * i));
                        s += (wv[0] * xv[0] + wv[1] * xv[1]) + (wv[2] * xv[2] + wv[3] * xv[3]);
                    }
                    lds[L::pz + q] = s;
                }
                __syncthreads();
                float w0 = 0.f, w1 = 0.f, w2 = 0.f;
                if (tid < 128) {
                    const float z0 = lds[L::zpart] + lds[L::zpart + 4] + lds[L::bsm], z1 = lds[L::zpart + 1] + lds[L::zpart + 5] + lds[L::bsm + 1],
                                z2 = lds[L::zpart + 2] + lds[L::zpart + 6] + lds[L::bsm + 2];
                    const float mx = fmaxf(z0, fmaxf(z1, z2));
                    const float e0 = __expf(z0 - mx), e1 = __expf(z1 - mx), e2 = __expf(z2 - mx);
                    const float inv = 1.f / (e0 + e1 + e2);
                    w0 = e0 * inv; w1 = e1 * inv; w2 = e2 * inv;
                    if (tid == 0) { float* m = a.MW + ((long long)t * B + r) * 3; m[0] = w0; m[1] = w1; m[2] = w2; }
                } else if (tid < 128 + 16) {                     // (P = 16) dropout, softmax over the classes: 16 lanes
                    const int j = tid - 128;
                    float s = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) s += lds[L::pz + j * 8 + cc];
                    const float zz = (s + lds[L::bsm + 4 + j]) * dropout_scale(sd, (uint64_t)r * P + j, a.p_pos, a.inv_pos);
                    a.DP[((long long)t * B + r) * P + j] = zz;
                    float m2 = zz;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 16));
                    const float e = __expf(zz - m2);
                    float sum = e;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
                    const float pw = e / sum;
                    lds[L::sz + j] = pw;
                    a.PW[((long long)t * B + r) * P + j] = pw;
                }
                __syncthreads();
                if (tid < 128) {                                 // attended feature: A / 128 values per thread, 16-byte pieces
                    const float* f = rows_lds + rs * L::per_row;
                    for (int k = 4 * tid; k < A; k += 512) {
                        const f32x4 v = w0 * *reinterpret_cast<const f32x4*>(f + k) + w1 * *reinterpret_cast<const f32x4*>(f + A + k) +
                                        w2 * *reinterpret_cast<const f32x4*>(f + 2 * A + k);
                        bstore4_sc1(rsAT, (unsigned)(r * A + k) * 4u, v);
                        bstore4_sc1(rsPA, (unsigned)(r * (R + A) + R + k) * 4u, v);
                    }
                } else {                                         // pos_feat = w_p2 . pw + b_p2   (w_p2 [R][P]: 64 contiguous bytes per output)
                    const int q = tid - 128;
                    for (int k = 4 * q; k < R; k += 512) {
                        f32x4 s = *reinterpret_cast<const f32x4*>(lds + L::bsm + 4 + P + k);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const f32x4* wr = reinterpret_cast<const f32x4*>(a.w_p2 + (long long)(k + e) * P);
#pragma unroll
                            for (int j4 = 0; j4 < P / 4; ++j4) {
                                const f32x4 wv = wr[j4], pv = *reinterpret_cast<const f32x4*>(lds + L::sz + 4 * j4);
                                s[e] += (wv[0] * pv[0] + wv[1] * pv[1]) + (wv[2] * pv[2] + wv[3] * pv[3]);
                            }
                        }
                        bstore4_sc1(rsPA, (unsigned)(r * (R + A) + k) * 4u, s);
                    }
                }
            }
        }
        stamp(c, 7);
        seam_arrive(a.sync + 1 * SHARDS * LINE, shard);
        stamp(c, 8);
        // ---- shadow of seam 1: the h_lang block of the language-LSTM gates, kept in LDS for phase E
        run_phase<NBR, 0, true>(c, TILES_U, rsWcat, (unsigned)((5 * R + G1 + gate_row_l) * R) * 4u, rsWcat, 0u, rsHL, rsHL, (unsigned)R * 4u, kwR, nullptr, 0u, 0u, -1,
                          nopre, [&](float v, float, int, int slot) { jobs_lds[slot * L::per_job + 256 + tid] = v; });
        stamp(c, 9);
        // =================== phase C: gate-MLP layer 1 ===================
        {
            const __amdgpu_buffer_rsrc_t rsPA = make_rsrc(a.PA + (long long)t * B * (R + A), (long long)B * (R + A) * 4);
            if (!run_phase<NBPA, 0, true>(c, TILES_G1, rsWg1, (unsigned)((16 * tG1 + l15) * G1 + R) * 4u, rsWg1, 0u, rsPA, rsPA, (unsigned)(R + A) * 4u, kwPA,
                                    a.sync + 1 * SHARDS * LINE, tgt, 0x100u + 1, 10, nopre, [&](float v, float, int rb, int slot) {
                                        const int n = 16 * tG1 + col, grow = rb * 16 + row;
                                        v += jobs_lds[slot * L::per_job + tid] + bG1;
                                        v = (v > 0.f ? v : 0.f) * dropout_scale(sg, (uint64_t)grow * G1 + n, a.p_lm, a.inv_lm);
                                        if (grow < B) store_sc1(a.G1D + ((long long)t * B + grow) * G1 + n, v);
                                    }))
                return;
        }
        stamp(c, 11);
        seam_arrive(a.sync + 2 * SHARDS * LINE, shard);
        // =================== phase D: gate-MLP layer 2, gate, gated feature ===================
        {
            const __amdgpu_buffer_rsrc_t rsG = make_rsrc(a.G1D + (long long)t * B * G1, (long long)B * G1 * 4);
            if (!run_phase<NBG, 0, true>(c, TILES_G2, rsWg2, (unsigned)((16 * tG2 + l15) * G1) * 4u, rsWg2, 0u, rsG, rsG, (unsigned)G1 * 4u, kwG,
                                   a.sync + 2 * SHARDS * LINE, tgt, 0x100u + 2, 12,
                                   [&](int rb, int) { return load_sc1(a.AT + ((long long)t * B + min(rb * 16 + row, B - 1)) * A + 16 * tG2 + col); },
                                   [&](float v, float att, int rb, int) {
                                       const int n = 16 * tG2 + col, grow = rb * 16 + row;
                                       if (grow < B) {
                                           const long long o = ((long long)t * B + grow) * A + n;
                                           const float gte = sigm(v + bG2);
                                           a.GT[o] = gte;
                                           store_sc1(a.GA + o, gte * att);
                                       }
                                   }))
                return;
        }
        stamp(c, 13);
        seam_arrive(a.sync + 3 * SHARDS * LINE, shard);
        // =================== phase E: language-LSTM gates + cell ===================
        {
            const __amdgpu_buffer_rsrc_t rsGA = make_rsrc(a.GA + (long long)t * B * A, (long long)B * A * 4);
            if (!run_phase<NBA, 0, true>(c, TILES_U, rsWil, (unsigned)(gate_row_l * ldw_l + a.We) * 4u, rsWil, 0u, rsGA, rsGA, (unsigned)A * 4u, kwA,
                                   a.sync + 3 * SHARDS * LINE, tgt, 0x100u + 3, 14,
                                   [&](int rb, int) { return a.GXL[((long long)t * B + min(rb * 16 + row, B - 1)) * 4 * R + gate * R + unit]; },
                                   [&](float v, float gx, int rb, int slot) {
                                       const int grow = rb * 16 + row;
                                       const float pre = gx + (jobs_lds[slot * L::per_job + 256 + tid] + v) + bE;
                                       const float act = gate == 2 ? tanhf(pre) : sigm(pre);
                                       if (grow < B) a.SL[((long long)t * B + grow) * 4 * R + gate * R + unit] = act;
                                       const float fg = __shfl_down(act, 4, 64), gg = __shfl_down(act, 8, 64), og = __shfl_down(act, 12, 64);
                                       if (gate == 0) {
                                           float* cs = jobs_lds + slot * L::per_job + 576 + row * 4 + (col & 3);
                                           const float cn = fg * cs[0] + act * gg;
                                           const float hn = og * tanhf(cn);
                                           cs[0] = cn;
                                           if (grow < B) {
                                               a.CL[(long long)(t + 1) * BR + (long long)grow * R + unit] = cn;
                                               store_sc1(HLn + (long long)grow * R + unit, hn);
                                           }
                                       }
                                   }))
                return;
        }
        stamp(c, 15);
        seam_arrive(a.sync + 4 * SHARDS * LINE, shard);
    }
}

constexpr int PR = 512, PA_ = 1024, PP = 16;

bool persistent_shape_ok(const ekaid_speaker_seq_t* d) {
    return d->R == PR && d->A == PA_ && d->P == PP && d->G1 == 2 * PR + PA_ && d->T >= 1 && d->T <= PMAX_T && d->B >= 1 && (d->We % 4) == 0;
}

struct PersistPlan { int G, nrb, jobs_max, rows_max; size_t lds_bytes; };

bool persistent_plan(const ekaid_speaker_seq_t* d, PersistPlan& pl) {
    if (!persistent_shape_ok(d)) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // TWO workgroups per CU (each at most half the registers and LDS: both are resident, the second one's matrix work covers the
    // first one's load waits), and a multiple of the 128 gate tiles so that a workgroup keeps ONE column tile for all its row blocks
    static const int wg_per_cu = [] { const char* e = getenv("EKAID_PERSIST_WG_PER_CU"); return e && e[0] == '2' ? 2 : 1; }();
    int G = cus >= 256 ? 256 * wg_per_cu : (cus >= 128 ? 128 * wg_per_cu : 0);
    if (!G) return false;
    pl.G = G;
    pl.nrb = (d->B + 15) / 16;
    pl.jobs_max = ((PR / 4) * pl.nrb + G - 1) / G;              // also bounds the 128 layer-1 tiles' slots (G1 / 16 == R / 4)
    pl.rows_max = (d->B + G - 1) / G;
    using L = Lds<PR, PA_, PP>;
    pl.lds_bytes = (size_t)(L::fixed + pl.jobs_max * L::per_job + pl.rows_max * L::per_row) * sizeof(float);
    return pl.lds_bytes <= 80 * 1024;
}

}  // namespace
}  // namespace ekaid

using namespace ekaid;

// EKAID_PERSIST_TRACE=1: every workgroup stamps 16 points of every step (s_memrealtime) behind the seam counters (tools/seq_bench.py)
static bool trace_enabled() {
    static const bool on = [] { const char* e = getenv("EKAID_PERSIST_TRACE"); return e && e[0] == '1'; }();
    return on;
}
static size_t sync_bytes() { return ((size_t)SYNC_WORDS * 4 + 255) & ~(size_t)255; }

extern "C" size_t ekaid_speaker_seq_persistent_workspace_bytes(const ekaid_speaker_seq_t* d) {
    PersistPlan pl;
    if (!d || !persistent_plan(d, pl)) return 0;
    return sync_bytes() + (trace_enabled() ? (size_t)pl.G * d->T * 16 * sizeof(unsigned long long) : 0);
}

extern "C" int ekaid_speaker_seq_fwd_persistent(const ekaid_speaker_seq_t* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d) { set_error("ekaid_speaker_seq_fwd_persistent: null descriptor"); return EKAID_ERR_ARG; }
    PersistPlan pl;
    if (!persistent_plan(d, pl)) {
        set_error("ekaid_speaker_seq_fwd_persistent: unsupported shape (R=%d A=%d P=%d G1=%d T=%d B=%d) or device", d->R, d->A, d->P, d->G1, d->T, d->B);
        return EKAID_ERR_UNSUPPORTED;
    }
    const size_t need = ekaid_speaker_seq_persistent_workspace_bytes(d);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        set_error("ekaid_speaker_seq_fwd_persistent: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
        return EKAID_ERR_ARG;
    }
    if (!d->seeds && (d->p_lm > 0.f || d->p_pos > 0.f)) { set_error("ekaid_speaker_seq_fwd_persistent: dropout without seeds"); return EKAID_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    PFwd a;
    memset(&a, 0, sizeof(a));
    a.T = d->T; a.B = d->B; a.G = pl.G; a.nrb = pl.nrb;
    a.p_lm = d->p_lm; a.p_pos = d->p_pos;
    a.inv_lm = d->p_lm > 0.f ? 1.f / (1.f - d->p_lm) : 1.f;
    a.inv_pos = d->p_pos > 0.f ? 1.f / (1.f - d->p_pos) : 1.f;
    a.ep = dropout_epoch_ptr();
    a.Wcat = d->Wcat; a.whh_a = d->whh_a; a.bhh_a = d->bhh_a; a.w_fc = d->w_fc; a.b_fc = d->b_fc; a.b_p1 = d->b_p1; a.w_wp = d->w_wp;
    a.b_wp = d->b_wp; a.w_p2 = d->w_p2; a.b_p2 = d->b_p2; a.w_g1 = d->w_g1; a.b_g1 = d->b_g1; a.w_g2 = d->w_g2; a.b_g2 = d->b_g2;
    a.wih_l = d->wih_l; a.bhh_l = d->bhh_l; a.bef = d->bef; a.diff = d->diff; a.aft = d->aft; a.GXA = d->GXA; a.GXL = d->GXL;
    a.HA = d->HA; a.CA = d->CA; a.HL = d->HL; a.CL = d->CL; a.SA = d->SA; a.SL = d->SL; a.G1D = d->G1D; a.MW = d->MW; a.VP = d->VP;
    a.DP = d->DP; a.PW = d->PW; a.PA = d->PA; a.AT = d->AT; a.GT = d->GT; a.GA = d->GA;
    a.sync = (unsigned*)workspace;
    a.trace = trace_enabled() ? reinterpret_cast<unsigned long long*>((char*)workspace + sync_bytes()) : nullptr;
    a.We = d->We; a.jobs_max = pl.jobs_max; a.rows_max = pl.rows_max;
    if (d->seeds) memcpy(a.seeds, d->seeds, sizeof(uint64_t) * 3 * d->T);
    const void* ptrs[] = {a.Wcat, a.whh_a, a.bhh_a, a.w_fc, a.b_fc, a.b_p1, a.w_wp, a.b_wp, a.w_p2, a.b_p2, a.w_g1, a.b_g1, a.w_g2, a.b_g2, a.wih_l,
                          a.bhh_l, a.bef, a.diff, a.aft, a.GXA, a.GXL, a.HA, a.CA, a.HL, a.CL, a.SA, a.SL, a.G1D, a.MW, a.VP, a.DP, a.PW, a.PA,
                          a.AT, a.GT, a.GA};
    for (const void* p : ptrs)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) { set_error("ekaid_speaker_seq_fwd_persistent: null or misaligned (16 B) pointer in the descriptor"); return EKAID_ERR_ARG; }
    auto kern = spk_seq_fwd_persistent_kernel<PR, PA_, PP>;
    if (func_attr_once(40)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("ekaid_speaker_seq_fwd_persistent: cannot raise the dynamic LDS limit");
            return EKAID_ERR_LAUNCH;
        }
    }
    // seam counters and the status word start at zero in EVERY launch (a memset node when captured)
    if (hipMemsetAsync(workspace, 0, sync_bytes(), st) != hipSuccess) { (void)hipGetLastError(); set_error("ekaid_speaker_seq_fwd_persistent: memset failed"); return EKAID_ERR_LAUNCH; }
    hipLaunchKernelGGL(kern, dim3(pl.G), dim3(256), pl.lds_bytes, st, a);
    return check_launch("speaker_seq_fwd_persistent");
}

/* status word of the last persistent launch on this workspace (device pointer to one unsigned; 0 = fine, 0x100 + seam = a seam
 * did not open in time and the launch was abandoned) */
extern "C" const unsigned* ekaid_speaker_seq_persistent_status_ptr(const void* workspace) {
    return workspace ? reinterpret_cast<const unsigned*>(workspace) + NSEAM * SHARDS * LINE : nullptr;
}
