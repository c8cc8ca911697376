// Persistent sequence kernels: the whole time loop of the teacher-forced answer decoder in ONE launch.
//
// ref: dynamic_speaker_change_pos.py:94-131 (DynamicCore.forward) inside :196-222 (the loop of DynamicSpeaker._forward).
//
// The step-wise path (csrc/sequence.hip) issues 8 dependent launches per decode step: four split-K slab products and four
// kernels that sum the slabs.  At 32-256 batch rows every one of them is latency-bound (7-14 us for < 1 us of arithmetic), and
// the chain is what the real-data loop, greedy decoding and the 8-GPU strong-scaling point spend their time on.  Here one
// workgroup per CU stays resident for all T steps:
//   * every product of a step is cut into jobs of 16 batch rows x (1, 2 or 4 tiles of) 16 output columns with the FULL contraction
//     (no split-K, no slabs: each output element is produced by one workgroup in a fixed order -> deterministic by construction);
//     the four waves of the workgroup split K, exact f32 arithmetic on v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain per wave),
//     the partial tiles summed through LDS.  A workgroup owns the same (column tiles, row blocks) in every step: its weight
//     fragments are fetched once per phase into registers BEFORE the wait they hide behind, LSTM cell states never leave LDS;
//   * what one phase hands to the next travels in FRAGMENT-MAJOR layout: the activation matrix of a row block is stored as 1-KB
//     chunks [k / 16][(k % 16) / 4][row][k % 4], exactly the order in which the 64 lanes of a wave hold an MFMA A operand, so every
//     load instruction of a consumer reads 1 KB contiguous (row-major rows cost 16 half-lines per instruction and ran at a fifth of
//     the CU's load rate).  The row-major tensors of the saved-tensor contract are written next to it (plain stores, nobody waits);
//   * the all-to-all volume through the fabric is what bounds a step (every column tile reads all rows): all products that read
//     h_lang_{t-1} / h_att_{t-1} run in ONE pass over those rows (attention-LSTM gates, POS layer 1, the h_lang blocks of gate-MLP
//     layer 1 and of the language LSTM); gate-MLP layer 1 is not a product at all -- its [pos_feat | att] input is linear in the
//     16 POS weights and the 3 module weights of a row, so with F_m = feat_m W_att^T (once per sequence), Wp = W_pp W_p2 and
//     cp = W_pp b_p2 (once per call) the pre-activation is mw . F + pw . Wp + cp + (h_lang block) + b: 19 multiply-adds per element
//     instead of a K = 1536 contraction; long contractions run 2 or 4 column tiles per job at large batch (activations read once
//     per job);
//   * the five dependencies of a step (h_att/vpos -> row-local module and POS weights -> gate-MLP layer 1 -> layer 2 -> language LSTM
//     -> next step) are device-scope seams instead of kernel boundaries: handed-off bytes are stored write-through (sc1), every
//     storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane adds to the seam's counter (sharded per XCD, each
//     shard on a line of its own); consumers poll the eight shards with sc1 loads from one wave, barrier, and read the handed-off
//     bytes ONLY with sc1 loads (cdna_hip_programming.md Guideline 16, form R1 / MI355X_MICROARCH.md "Valid forms", first table row).
//     Counters count up over the steps of one launch and are zeroed by a memset node in front of every launch.  A hand-off buffer is
//     rewritten only after a later seam has proven every reader of the previous contents done (h_att is double-buffered: it is read
//     and rewritten in the same phase).  Spins are bounded: a seam that does not open within ~0.5 s sets the status word and every
//     workgroup leaves.
// The saved-tensor contract is that of ekaid_speaker_seq_fwd (include/ekaid_hip.h): either forward pairs with either backward.
#include "../../include/ekaid_hip.h"
#include "persist.h"

namespace ekaid {
namespace {

struct PFwd {
    int T, B, G, nrb, mtD, mtE;
    float p_lm, p_pos, inv_lm, inv_pos;
    const unsigned long long* ep;
    const float *Wcat, *whh_a, *bhh_a, *w_fc, *b_fc, *b_p1, *w_wp, *b_wp, *w_p2, *b_p2, *w_g1, *b_g1, *w_g2, *b_g2, *wih_l, *bhh_l;
    const float *bef, *diff, *aft, *GXA, *GXL;
    float *HA, *CA, *HL, *CL, *SA, *SL, *G1D, *MW, *VP, *DP, *PW, *PA, *AT, *GT, *GA;
    unsigned* sync;
    unsigned* status;                 // per-device sticky word: non-zero = a seam timed out (api.hip: seq_status_ptr)
    unsigned long long* trace;        // NULL or [G][T][16] stamps (s_memrealtime, 100 MHz)
    // workspace: hand-off buffers (fragment-major unless noted) and the per-call linear maps of gate-MLP layer 1
    float *HLfm, *HAfm /* [2] */, *G1fm, *GAfm, *MWPW /* [Bp][32] row-major: mw 0..2, pw 16..31 */, *LHP /* [nrb][R/4][256] */;
    const float *Fm /* [3][B][G1] */, *Wp /* [G1][P] */, *cp /* [G1] */;
    int We, jobsA, jobsD, jobsE, wp_lds;
    // greedy decoding (template flag GREEDY): the step loop additionally forms the logits, picks the tokens and feeds them back
    int V, seq_len, t_base;
    const float *emb, *bih_l, *w_lo, *b_lo;
    long long* it; int* unfinished; long long* seq; float* seq_logp; int* n_unfinished;
    float *XTfm /* [Bp][512]: relu(emb[token]) padded with zeros, fragment-major */, *LOG /* [Bp][160] logits, row-major */;
    uint64_t seeds[3 * PMAX_T];
};

// LDS carve (floats); every offset a multiple of 4 floats
template <int R, int A, int P>
struct Lds {
    static constexpr int flag = 0;                              // 4
    static constexpr int red = 4;                               // 2 x (4 tiles x 1280)
    static constexpr int hrow = red + 2 * 5120;                 // R   (h_att row of phase B)
    static constexpr int vrow = hrow + R;                       // R   (vpos row)
    static constexpr int zpart = vrow + R;                      // 8   (module-weight partials: 2 waves x 3)
    static constexpr int pz = zpart + 8;                        // 128 (POS logit partials)
    static constexpr int sz = pz + 128;                         // 64  (POS weights)
    static constexpr int wfc = sz + 64;                         // 3 R
    static constexpr int bsm = wfc + 3 * R;                     // b_fc (4) | b_wp (P) | b_p2 (R)
    static constexpr int stage = bsm + 4 + P + R;               // MAXJ x 512: [mw | pw] rows of the row blocks of this workgroup's jobs
    static constexpr int fixed = stage + MAXJ * 512;
    // then per phase-A job slot: g1p 256 | catt 64 | F slices 3 x 256 ; then per phase-D (job, tile) slot: feature slices 3 x 256;
    // then per phase-E (job, tile) slot: clang 64
    static constexpr int per_jobA = 256 + 64 + 768;
    static constexpr int per_tileD = 768;
    static constexpr int per_tileE = 64 + 256;                  // language-LSTM cell state | (greedy) word-side gates of the tile
};

struct Pre3 { float a, b, c; };
struct PreE { float gx[4], lh[4]; };

template <int R, int A, int P, int MTD, int MTE, bool GREEDY>
__global__ __launch_bounds__(256, 1) void spk_seq_fwd_persistent_kernel(const PFwd a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = Lds<R, A, P>;
    constexpr int G1 = 2 * R + A, NW = 9 * R + G1;
    constexpr int NBR = R / 64, NBA = A / 64, NBG = G1 / 64;   // 16-k blocks per wave of a K-long contraction
    constexpr int KBR = R / 16, KBA = A / 16, KBG = G1 / 16;    // chunks per row block of the fragment-major buffers
    constexpr int TILES_U = R / 4;                               // gate tiles: 4 hidden units x 4 gates  (128)
    constexpr int TILES_P1 = R / 16, TILES_G1 = G1 / 16, TILES_G2 = A / 16;
    static_assert(TILES_G1 == TILES_U, "layer-1 tiles and gate tiles share their owner");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int B = a.B, G = a.G, nrb = a.nrb;
    const int l15 = lane & 15, kq4 = (lane >> 4) * 4;
    const int row = tid >> 4, col = tid & 15;                    // element of a reduced tile
    const unsigned shard = xcc_id();
    unsigned* const status = a.status;
    float* const jobsA_lds = lds + L::fixed;
    float* const tilesD_lds = jobsA_lds + a.jobsA * L::per_jobA;
    float* const tilesE_lds = tilesD_lds + a.jobsD * MTD * L::per_tileD;
    float* const wwp_lds = tilesE_lds + a.jobsE * MTE * L::per_tileE;     // (a.wp_lds) weight_pos, rows padded to R + 4 floats
    Ctx c = {tid, lane, wave, wg, G, nrb, B, row, col, lds + L::red, 0, status, reinterpret_cast<int*>(lds + L::flag), nullptr};
    const long long BR = (long long)B * R;
    const long long Bp = 16ll * nrb;

    // ---- ownership.  Phase A / C: gate tile tU = wg % 128 (4 hidden units; also layer-1 column tile tU and, for tU < 32, POS tile tU),
    // row blocks rbA0 + n drbA.  Phase D: tile group tD of MTD layer-2 tiles.  Phase E: tile group tE of MTE gate tiles.
    const int tU = wg % TILES_U, rbA0 = wg / TILES_U, drbA = G / TILES_U;
    const int njA = wg < TILES_U * nrb ? (TILES_U * nrb - wg + G - 1) / G : 0;
    const bool hasP1 = tU < TILES_P1;
    constexpr int TG_D = TILES_G2 / MTD, TG_E = TILES_U / MTE;
    const int tD = wg % TG_D, rbD0 = wg / TG_D, drbD = G / TG_D;
    const int njD = wg < TG_D * nrb ? (TG_D * nrb - wg + G - 1) / G : 0;
    const int tE = wg % TG_E, rbE0 = wg / TG_E, drbE = G / TG_E;
    const int njE = wg < TG_E * nrb ? (TG_E * nrb - wg + G - 1) / G : 0;
    const int gate = col >> 2, ucol = col & 3;
    const int gate_row_l = (l15 >> 2) * R + (l15 & 3);           // + 4 tile: weight row of lane l15 in a gate tile (4 units x 4 gates)

    // ---- one-time preloads
    for (int i = tid; i < 3 * R; i += 256) lds[L::wfc + i] = a.w_fc[i];
    if (tid < 3) lds[L::bsm + tid] = a.b_fc[tid];
    if (tid < P) lds[L::bsm + 4 + tid] = a.b_wp[tid];
    for (int i = tid; i < R; i += 256) lds[L::bsm + 4 + P + i] = a.b_p2[i];
    if (a.wp_lds && wg < B)                                       // (only the workgroups that own a batch row read it)
        for (int i = tid; i < P * R; i += 256) wwp_lds[(i / R) * (R + 4) + i % R] = a.w_wp[i];
    for (int n = 0; n < njA; ++n) {                              // attention-LSTM cell state and the F_m slices of the (tile, row block) pairs
        const int rb = rbA0 + n * drbA, gr = min(rb * 16 + row, B - 1);
        float* s = jobsA_lds + n * L::per_jobA;
        if (tid < 64) s[256 + tid] = a.CA[(long long)min(rb * 16 + (tid >> 2), B - 1) * R + 4 * tU + (tid & 3)];
#pragma unroll
        for (int m = 0; m < 3; ++m) s[320 + m * 256 + tid] = a.Fm[((long long)m * B + gr) * G1 + 16 * tU + col];
    }
    for (int n = 0; n < njD; ++n) {                              // feature slices of the layer-2 tiles (attended feature = mw . these)
        const int rb = rbD0 + n * drbD, gr = min(rb * 16 + row, B - 1);
#pragma unroll
        for (int m = 0; m < MTD; ++m) {
            float* s = tilesD_lds + (n * MTD + m) * L::per_tileD;
            const long long o = (long long)gr * A + 16 * (tD * MTD + m) + col;
            s[tid] = a.bef[o]; s[256 + tid] = a.diff[o]; s[512 + tid] = a.aft[o];
        }
    }
    for (int n = 0; n < njE; ++n) {                              // language-LSTM cell state
        const int rb = rbE0 + n * drbE;
#pragma unroll
        for (int m = 0; m < MTE; ++m)
            if (tid < 64) tilesE_lds[(n * MTE + m) * L::per_tileE + tid] = a.CL[(long long)min(rb * 16 + (tid >> 2), B - 1) * R + 4 * (tE * MTE + m) + (tid & 3)];
    }
    // biases and per-column constants of the owned tiles stay in registers for the whole sequence
    const float bA = a.bhh_a[gate * R + 4 * tU + ucol], bP1 = hasP1 ? a.b_p1[16 * tU + col] : 0.f;
    const float bC = a.b_g1[16 * tU + col] + a.cp[16 * tU + col];
    float wp[P];
#pragma unroll
    for (int j = 0; j < P; ++j) wp[j] = a.Wp[(long long)(16 * tU + col) * P + j];
    float bD[MTD], bE[MTE];
#pragma unroll
    for (int m = 0; m < MTD; ++m) bD[m] = a.b_g2[16 * (tD * MTD + m) + col];
#pragma unroll
    for (int m = 0; m < MTE; ++m) bE[m] = a.bhh_l[gate * R + 4 * (tE * MTE + m) + ucol];
    float bIh[MTE];
#pragma unroll
    for (int m = 0; m < MTE; ++m) bIh[m] = GREEDY ? a.bih_l[gate * R + 4 * (tE * MTE + m) + ucol] : 0.f;

    const __amdgpu_buffer_rsrc_t rsWcat = make_rsrc(a.Wcat, (long long)NW * R * 4), rsWhh = make_rsrc(a.whh_a, 4ll * R * R * 4),
                                 rsWg2 = make_rsrc(a.w_g2, (long long)A * G1 * 4), rsWil = make_rsrc(a.wih_l, 4ll * R * (a.We + A) * 4);
    const __amdgpu_buffer_rsrc_t rsHLfm = make_rsrc(a.HLfm, Bp * R * 4), rsG1fm = make_rsrc(a.G1fm, Bp * G1 * 4), rsGAfm = make_rsrc(a.GAfm, Bp * A * 4),
                                 rsMWPW = make_rsrc(a.MWPW, Bp * 32 * 4);
    const unsigned kwR = (unsigned)(wave * (R / 4) + kq4) * 4u;          // byte offset of this lane's first k in a K = R contraction
    const unsigned kwA = (unsigned)(wave * (A / 4) + kq4) * 4u;
    const unsigned kwG = (unsigned)(wave * (G1 / 4) + kq4) * 4u;
    const int ldw_l = a.We + A;
    const unsigned lo = (unsigned)lane * 16u;

    // ---- prologue: the initial states into fragment-major layout (chunk per workgroup pass), then the first seam
    {
        const int nchunk = nrb * KBR;
        const int kq = tid >> 6, rr = (tid >> 2) & 15, e = tid & 3;
        for (int ch = wg; ch < 2 * nchunk; ch += G) {
            const int which = ch >= nchunk, cc = ch - which * nchunk, rb = cc / KBR, kb = cc - rb * KBR, gr = rb * 16 + rr;
            const float* src = which ? a.HA : a.HL;
            const float v = gr < B ? src[(long long)gr * R + kb * 16 + kq * 4 + e] : 0.f;
            store_sc1((which ? a.HAfm : a.HLfm) + ((long long)cc << 8) + tid, v);
        }
        if (GREEDY) {                                            // word input of the first step: relu(emb[token]) of the tokens handed in
            const __amdgpu_buffer_rsrc_t rsXT = make_rsrc(a.XTfm, Bp * 512 * 4);
            for (int r = wg; r < B; r += G) {
                if (tid < 128) {
                    const long long tok = a.it[r];
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (4 * tid < a.We) {
                        x = *reinterpret_cast<const f32x4*>(a.emb + tok * a.We + 4 * tid);
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) x[e2] = x[e2] > 0.f ? x[e2] : 0.f;
                    }
                    bstore4_sc1(rsXT, fm_chunk(r >> 4, tid >> 2, 32) + (unsigned)(((tid & 3) * 16 + (r & 15)) * 16), x);
                }
            }
        }
        seam_arrive(a.sync + 5 * SHARDS * LINE, shard);
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t rsXTfm = make_rsrc(a.XTfm, Bp * 512 * 4), rsWlo = make_rsrc(a.w_lo, (long long)(GREEDY ? a.V : 1) * R * 4);

    // ---- greedy decoding: the two stations that turn h_lang_t into the next token.  They run INSIDE step t + 1 and ride on its seams:
    // the logits between the arrival at seam 0 and its wait (h_lang_t stays in its hand-off buffer until phase E of step t + 1; every
    // workgroup has waited for it at the top of phase A); the row-local part behind the wait of seam 1 -- a workgroup arrives there after
    // its logits, so seam 1 also proves every logit stored; the word input of phase E behind the wait of seam 2, for the same reason.  No
    // seam of their own; the last step's run after the loop, with one.
    auto logits_station = [&](const unsigned* seam4, unsigned target4) -> bool {
        // logits = h_lang . W_logit^T + b (V columns = ceil(V / 16) tiles in 16 tile slots), eval mode: no output dropout
        const int tL = wg % 16, vcol = 16 * tL + col;
        const float blo = vcol < a.V ? a.b_lo[vcol] : 0.f;
        if (!run_mt<NBR, 1, PreE>(
                c, 16, [&](int, int i) { return bload4<0>(rsWlo, (unsigned)(min(16 * tL + l15, a.V - 1) * R) * 4u + kwR + 64u * i); }, rsHLfm, KBR,
                seam4, target4, 0x100u + 4, -1, [&](int, int) { return PreE{}; },
                [&](const float (&v)[1], const PreE&, int rb, int) {
                    const int grow = rb * 16 + row;
                    if (grow < B && vcol < a.V) store_sc1(a.LOG + (long long)grow * 160 + vcol, v[0] + blo);
                }))
            return false;
        return true;
    };
    auto rows_station = [&](int tp) -> bool {
        if (wg >= B) return true;                                // (one row per workgroup and pass)
        // row-local: log-softmax, arg-max (lowest index on ties, token 0 forbidden at the first step), stop flags, next token
        const int ta = a.t_base + tp;
        for (int r = wg; r < B; r += G) {
            if (tid < 64) {
                float x[3], mx = -INFINITY;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int v = lane + 64 * q;
                    x[q] = v < a.V ? load_sc1(a.LOG + (long long)r * 160 + v) : -INFINITY;
                    mx = fmaxf(mx, x[q]);
                }
                mx = wave_max(mx);
                float sm = 0.f;
#pragma unroll
                for (int q = 0; q < 3; ++q) sm += lane + 64 * q < a.V ? expf(x[q] - mx) : 0.f;
                sm = wave_sum(sm);
                const float ls = logf(sm);
                float best = -INFINITY;
                int arg = a.V;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int v = lane + 64 * q;
                    float lp = v < a.V ? (x[q] - mx) - ls : -INFINITY;
                    if (ta == 0 && v == 0) lp = -INFINITY;
                    if (lp > best) { best = lp; arg = v; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ob = __shfl_xor(best, off, 64);
                    const int oa = __shfl_xor(arg, off, 64);
                    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
                }
                if (lane == 0) {
                    const int un = ta == 0 ? (arg > 0) : (a.unfinished[r] && arg > 0);
                    a.unfinished[r] = un;
                    const long long tok = un ? arg : 0;
                    a.it[r] = tok;
                    a.seq[(long long)r * a.seq_len + ta] = tok;
                    a.seq_logp[(long long)r * a.seq_len + ta] = best;
                    if (un) atomicAdd(a.n_unfinished + ta, 1);
                    reinterpret_cast<int*>(lds + L::zpart)[0] = (int)tok;
                }
            }
            __syncthreads();
            if (tid < 128) {                                 // relu(emb[token]) -> the word input of the next step, fragment-major
                const long long tok = reinterpret_cast<const int*>(lds + L::zpart)[0];
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (4 * tid < a.We) {
                    x = *reinterpret_cast<const f32x4*>(a.emb + tok * a.We + 4 * tid);
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) x[e2] = x[e2] > 0.f ? x[e2] : 0.f;
                }
                bstore4_sc1(rsXTfm, fm_chunk(r >> 4, tid >> 2, 32) + (unsigned)(((tid & 3) * 16 + (r & 15)) * 16), x);
            }
            __syncthreads();
        }
        return true;
    };

    for (int t = 0; t < a.T; ++t) {
        float* HAn = a.HA + (long long)(t + 1) * BR;
        float* HLn = a.HL + (long long)(t + 1) * BR;
        float* HAfm_r = a.HAfm + (long long)(t & 1) * Bp * R;          // h_att_{t-1}
        float* HAfm_w = a.HAfm + (long long)((t + 1) & 1) * Bp * R;    // h_att_t
        const __amdgpu_buffer_rsrc_t rsHAr = make_rsrc(HAfm_r, Bp * R * 4), rsHAw = make_rsrc(HAfm_w, Bp * R * 4);
        const DropSeed sv = {a.seeds[3 * t], a.ep}, sd = {a.seeds[3 * t + 1], a.ep}, sg = {a.seeds[3 * t + 2], a.ep};
        const unsigned tgt = (unsigned)G * (unsigned)(t + 1);
        const int ts = GREEDY ? 0 : t;                            // greedy decoding: the activation buffers are ONE step large
        c.trace = a.trace ? a.trace + ((long long)wg * a.T + t) * 16 : nullptr;
        stamp(c, 0);

        // =================== phase A: everything that reads h_lang_{t-1} / h_att_{t-1}, ONE pass over the rows ===================
        // attention-LSTM gates (+ cell), POS layer 1, the h_lang blocks of gate-MLP layer 1 (kept in LDS for phase C) and of the language-
        // LSTM gates (handed to the owner of phase E through LHP)
        {
            f32x4 wAh[NBR], wAa[NBR], w3[NBR], w4[NBR], w2[NBR], x[NBR];
            float gxa_next = 0.f;
            auto gxa_of = [&](int rb) { return a.GXA[((long long)ts * B + min(rb * 16 + row, B - 1)) * 4 * R + gate * R + 4 * tU + ucol]; };
            if (njA) {
                const unsigned woA = (unsigned)((gate_row_l + 4 * tU) * R) * 4u + kwR;
#pragma unroll
                for (int i = 0; i < NBR; ++i) {
                    wAh[i] = bload4<0>(rsWcat, woA + 64u * i);
                    wAa[i] = bload4<0>(rsWhh, woA + 64u * i);
                    w3[i] = bload4<0>(rsWcat, (unsigned)((5 * R + 16 * tU + l15) * R) * 4u + kwR + 64u * i);
                    w4[i] = bload4<0>(rsWcat, (unsigned)((5 * R + G1 + gate_row_l + 4 * tU) * R) * 4u + kwR + 64u * i);
                    w2[i] = hasP1 ? bload4<0>(rsWcat, (unsigned)((4 * R + 16 * tU + l15) * R) * 4u + kwR + 64u * i) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                gxa_next = gxa_of(rbA0);
            }
            if (!seam_wait(t > 0 ? a.sync + 4 * SHARDS * LINE : a.sync + 5 * SHARDS * LINE, t > 0 ? (unsigned)G * (unsigned)t : (unsigned)G, status,
                           c.ok_lds, 0x100u + 4))
                return;
            stamp(c, 1);
            if (njA) {
                unsigned ao = fm_chunk(rbA0, wave * NBR, KBR) + lo;
#pragma unroll
                for (int i = 0; i < NBR; ++i) x[i] = bload4<AUX_SC1>(rsHLfm, ao + 1024u * i);
#pragma unroll 1
                for (int n = 0; n < njA; ++n) {
                    {
                        const int rb = rbA0 + n * drbA;
                        const bool more = n + 1 < njA;
                        const unsigned ao2 = fm_chunk(rb + drbA, wave * NBR, KBR) + lo;
                        const float gxa = gxa_next;
                        if (more) gxa_next = gxa_of(rb + drbA);
                        f32x4 accA[2] = {}, acc3 = {}, acc4 = {}, acc2 = {};
                        // h_lang blocks: four products share them; the ring slot is refilled with the h_att block of the same index
#pragma unroll
                        for (int i = 0; i < NBR; ++i) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                accA[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], wAh[i][e], accA[0], 0, 0, 0);
                                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], w3[i][e], acc3, 0, 0, 0);
                                acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], w4[i][e], acc4, 0, 0, 0);
                                if (hasP1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], w2[i][e], acc2, 0, 0, 0);
                            }
                            x[i] = bload4<AUX_SC1>(rsHAr, ao + 1024u * i);
                        }
                        // h_att blocks: the attention-LSTM's own recurrence; refilled with the next job's h_lang blocks
#pragma unroll
                        for (int i = 0; i < NBR; ++i) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) accA[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i][e], wAa[i][e], accA[e & 1], 0, 0, 0);
                            if (more) x[i] = bload4<AUX_SC1>(rsHLfm, ao2 + 1024u * i);
                        }
                        ao = ao2;
                        f32x4 part[4] = {accA[0] + accA[1], acc3, acc4, acc2};
                        float v[4];
                        tiles_reduce<4>(part, v, c.red + c.par * 5120, wave, lane, tid);
                        c.par ^= 1;
                        const int grow = rb * 16 + row;
                        float* slot = jobsA_lds + n * L::per_jobA;
                        // (1) attention-LSTM cell: thread (row, col = gate * 4 + unit)
                        {
                            const int unit = 4 * tU + ucol;
                            const float pre = v[0] + gxa + bA;
                            const float act = gate == 2 ? tanhf(pre) : sigm(pre);
                            if (grow < B) a.SA[((long long)ts * B + grow) * 4 * R + gate * R + unit] = act;
                            const float fg = __shfl_down(act, 4, 64), gg = __shfl_down(act, 8, 64), og = __shfl_down(act, 12, 64);
                            float hn = 0.f;
                            if (gate == 0) {
                                float* cs = slot + 256 + row * 4 + ucol;
                                const float cn = fg * cs[0] + act * gg;
                                hn = og * tanhf(cn);
                                cs[0] = cn;
                                if (grow < B) {
                                    a.CA[(long long)(t + 1) * BR + (long long)grow * R + unit] = cn;
                                    HAn[(long long)grow * R + unit] = hn;
                                }
                            }
                            const f32x4 h4 = gather4(hn);           // lanes col 0..3 -> lane col 0: the 4 units of this row, 16 bytes of the hand-off chunk
                            if (col == 0) bstore4_sc1(rsHAw, fm_chunk(rb, tU >> 2, KBR) + (unsigned)(((tU & 3) * 16 + row) * 16), h4);
                        }
                        // (2) h_lang block of gate-MLP layer 1: stays with this workgroup
                        slot[tid] = v[1];
                        // (3) h_lang block of the language-LSTM gates: to the owner of the gate tile in phase E (chunk of 256, thread-indexed)
                        store_sc1(a.LHP + (((long long)rb * TILES_U + tU) << 8) + tid, v[2]);
                        // (4) POS layer 1: vpos = dropout(relu(. + b))
                        if (hasP1) {
                            const int k = 16 * tU + col;
                            float y = v[3] + bP1;
                            y = (y > 0.f ? y : 0.f) * dropout_scale(sv, (uint64_t)grow * R + k, a.p_lm, a.inv_lm);
                            if (grow < B) store_sc1(a.VP + ((long long)ts * B + grow) * R + k, y);
                        }
                    }
                }
            }
        }
        stamp(c, 2);
        seam_arrive(a.sync + 0 * SHARDS * LINE, shard);
        if (GREEDY && t > 0) {                                   // logits of step t - 1 (every workgroup has waited for seam 4 of that step above)
            if (!logits_station(nullptr, 0u)) return;
        }
        stamp(c, 3);
        if (!seam_wait(a.sync + 0 * SHARDS * LINE, tgt, status, c.ok_lds, 0x100u + 0)) return;
        stamp(c, 4);

        // =================== phase B: row-local module weights and POS weights ===================
        {
            const __amdgpu_buffer_rsrc_t rv = make_rsrc(a.VP + (long long)ts * BR, BR * 4);
            int rs = 0;
            for (int r = wg; r < B; r += G, ++rs) {
                if (rs) __syncthreads();
                if (tid < 128) *reinterpret_cast<f32x4*>(lds + L::hrow + 4 * tid) =
                    bload4<AUX_SC1>(rsHAw, fm_chunk(r >> 4, tid >> 2, KBR) + (unsigned)(((tid & 3) * 16 + (r & 15)) * 16));
                else *reinterpret_cast<f32x4*>(lds + L::vrow + 4 * (tid - 128)) = bload4<AUX_SC1>(rv, (unsigned)(r * R + 4 * (tid - 128)) * 4u);
                __syncthreads();
                if (tid < 128) {                                 // module-weight logits: 3 dot products over R
                    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
                    for (int k = tid; k < R; k += 128) {
                        const float xx = lds[L::hrow + k];
                        z0 += xx * lds[L::wfc + k]; z1 += xx * lds[L::wfc + R + k]; z2 += xx * lds[L::wfc + 2 * R + k];
                    }
                    z0 = wave_sum(z0); z1 = wave_sum(z1); z2 = wave_sum(z2);
                    if (lane == 0) { lds[L::zpart + wave * 4] = z0; lds[L::zpart + wave * 4 + 1] = z1; lds[L::zpart + wave * 4 + 2] = z2; }
                } else {                                         // POS logits: thread (j = q >> 3, c = q & 7) sums the k with (k / 4) % 8 == c
                    const int q = tid - 128, j = q >> 3, cq = q & 7;
                    // weight_pos row j from LDS when it fits (from L2 it is 16 dependent-latency loads per thread and row: 2.8 of the
                    // phase's 6.3 us), same order of operations either way
                    float s = 0.f;
                    if (a.wp_lds) {
                        const float* wl = wwp_lds + j * (R + 4) + 4 * cq;
#pragma unroll 4
                        for (int i = 0; i < R / 32; ++i) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + 32 * i), xv = *reinterpret_cast<const f32x4*>(lds + L::vrow + 4 * (cq + 8 * i));
                            s += (wv[0] * xv[0] + wv[1] * xv[1]) + (wv[2] * xv[2] + wv[3] * xv[3]);
                        }
                    } else {
                        const f32x4* wr = reinterpret_cast<const f32x4*>(a.w_wp + j * R) + cq;
#pragma unroll 4
                        for (int i = 0; i < R / 32; ++i) {
                            const f32x4 wv = wr[8 * i], xv = *reinterpret_cast<const f32x4*>(lds + L::vrow + 4 * (cq + 8 * i));
                            s += (wv[0] * xv[0] + wv[1] * xv[1]) + (wv[2] * xv[2] + wv[3] * xv[3]);
                        }
                    }
                    lds[L::pz + q] = s;
                }
                __syncthreads();
                if (tid < 3) {
                    const float z0 = lds[L::zpart] + lds[L::zpart + 4] + lds[L::bsm], z1 = lds[L::zpart + 1] + lds[L::zpart + 5] + lds[L::bsm + 1],
                                z2 = lds[L::zpart + 2] + lds[L::zpart + 6] + lds[L::bsm + 2];
                    const float mx = fmaxf(z0, fmaxf(z1, z2));
                    const float e0 = __expf(z0 - mx), e1 = __expf(z1 - mx), e2 = __expf(z2 - mx);
                    const float inv = 1.f / (e0 + e1 + e2);
                    const float wv = (tid == 0 ? e0 : (tid == 1 ? e1 : e2)) * inv;
                    a.MW[((long long)t * B + r) * 3 + tid] = wv;
                    store_sc1(a.MWPW + (long long)r * 32 + tid, wv);
                } else if (tid >= 128 && tid < 128 + 16) {       // (P = 16) dropout, softmax over the classes: 16 lanes
                    const int j = tid - 128;
                    float s = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) s += lds[L::pz + j * 8 + cc];
                    const float zz = (s + lds[L::bsm + 4 + j]) * dropout_scale(sd, (uint64_t)r * P + j, a.p_pos, a.inv_pos);
                    a.DP[((long long)ts * B + r) * P + j] = zz;
                    float m2 = zz;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 16));
                    const float ee = __expf(zz - m2);
                    float sum = ee;
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
                    const float pw = ee / sum;
                    lds[L::sz + j] = pw;
                    a.PW[((long long)ts * B + r) * P + j] = pw;
                    store_sc1(a.MWPW + (long long)r * 32 + 16 + j, pw);
                }
            }
        }
        stamp(c, 5);
        seam_arrive(a.sync + 1 * SHARDS * LINE, shard);
        stamp(c, 6);
        // ---- shadow of seam 1: pos_feat = w_p2 . pw + b_p2 of the rows just done -- only the saved-tensor contract reads it (PA[:, :R])
        {
            int rs = 0;
            for (int r = wg; r < B; r += G, ++rs) {
                if (rs == 0 && r + G >= B) {                     // the POS weights of the (single) row are still in LDS
                    if (tid >= 128) {
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
                            *reinterpret_cast<f32x4*>(a.PA + ((long long)ts * B + r) * (R + A) + k) = s;
                        }
                    }
                } else {                                         // several rows per workgroup: from the stored POS weights
                    for (int k = tid; k < R; k += 256) {
                        float s = lds[L::bsm + 4 + P + k];
                        const float* pwr = a.PW + ((long long)ts * B + r) * P;
#pragma unroll
                        for (int j = 0; j < P; ++j) s += a.w_p2[(long long)k * P + j] * pwr[j];
                        a.PA[((long long)ts * B + r) * (R + A) + k] = s;
                    }
                }
            }
        }
        // =================== phase C: gate-MLP layer 1 = mw . F + pw . Wp + (h_lang block) + constants; no contraction ===================
        if (!seam_wait(a.sync + 1 * SHARDS * LINE, tgt, status, c.ok_lds, 0x100u + 1)) return;
        stamp(c, 7);
        if (GREEDY && t > 0) {                                   // tokens of step t - 1 -> the word input phase E of this step reads behind seam 2
            if (!rows_station(t - 1)) return;
        }
        if (njA) {
            // [mw | pw] of the 16 rows of every owned row block -> LDS (128 x 16 bytes per block), one barrier
            for (int n = 0; n < njA; ++n)
                if (tid < 128)
                    *reinterpret_cast<f32x4*>(lds + L::stage + n * 512 + 4 * tid) =
                        bload4<AUX_SC1>(rsMWPW, (unsigned)(((rbA0 + n * drbA) * 16 + (tid >> 3)) * 32 + 4 * (tid & 7)) * 4u);
            __syncthreads();
            for (int n = 0; n < njA; ++n) {
                const int rb = rbA0 + n * drbA, grow = rb * 16 + row, nn = 16 * tU + col;
                const float* slot = jobsA_lds + n * L::per_jobA;
                const float* mp = lds + L::stage + n * 512 + row * 32;
                float y = slot[tid] + bC;
                y += mp[0] * slot[320 + tid] + mp[1] * slot[576 + tid] + mp[2] * slot[832 + tid];
#pragma unroll
                for (int j = 0; j < P; ++j) y += mp[16 + j] * wp[j];
                y = (y > 0.f ? y : 0.f) * dropout_scale(sg, (uint64_t)grow * G1 + nn, a.p_lm, a.inv_lm);
                if (grow < B) a.G1D[((long long)ts * B + grow) * G1 + nn] = y;
                const f32x4 y4 = gather4(y);
                if ((col & 3) == 0) bstore4_sc1(rsG1fm, fm_chunk(rb, tU, KBG) + (unsigned)(((col >> 2) * 16 + row) * 16), y4);
            }
        }
        stamp(c, 8);
        seam_arrive(a.sync + 2 * SHARDS * LINE, shard);
        stamp(c, 9);
        // =================== phase D: gate-MLP layer 2, gate, attended and gated feature ===================
        if (!run_mt<NBG, MTD, Pre3>(
                c, TG_D, [&](int m, int i) { return bload4<0>(rsWg2, (unsigned)((16 * (tD * MTD + m) + l15) * G1) * 4u + kwG + 64u * i); }, rsG1fm, KBG, a.sync + 2 * SHARDS * LINE, tgt,
                0x100u + 2, 10,
                [&](int rb, int) {                               // module weights of this thread's row (handed off at seam 1)
                    const float* mp = a.MWPW + (long long)(rb * 16 + row) * 32;
                    return Pre3{load_sc1(mp), load_sc1(mp + 1), load_sc1(mp + 2)};
                },
                [&](const float (&v)[MTD], const Pre3& mw, int rb, int n) {
                    const int grow = rb * 16 + row;
#pragma unroll
                    for (int m = 0; m < MTD; ++m) {
                        const int tau = tD * MTD + m, nn = 16 * tau + col;
                        const float* f = tilesD_lds + (n * MTD + m) * L::per_tileD;
                        const float att = mw.a * f[tid] + mw.b * f[256 + tid] + mw.c * f[512 + tid];
                        const float gte = sigm(v[m] + bD[m]);
                        const float ga = gte * att;
                        if (grow < B) {
                            const long long o = ((long long)ts * B + grow) * A + nn;
                            a.GT[o] = gte;
                            a.GA[o] = ga;
                            a.AT[o] = att;
                            a.PA[((long long)ts * B + grow) * (R + A) + R + nn] = att;
                        }
                        const f32x4 g4 = gather4(ga);
                        if ((col & 3) == 0) bstore4_sc1(rsGAfm, fm_chunk(rb, tau, KBA) + (unsigned)(((col >> 2) * 16 + row) * 16), g4);
                    }
                }))
            return;
        stamp(c, 11);
        seam_arrive(a.sync + 3 * SHARDS * LINE, shard);
        stamp(c, 12);
        if (GREEDY) {
            // word-side gates of the language LSTM, relu(emb[token]) . W_ih[:, :We]^T (K = We padded with zero activations to 512), by the
            // owners of phase E, kept in LDS; the tokens were stored before their writers arrived at seam 2 (prologue: seam 5)
            if (!run_mt<NBR, MTE, PreE>(
                    c, TG_E, [&](int m, int i) { return bload4<0>(rsWil, (unsigned)((gate_row_l + 4 * (tE * MTE + m)) * ldw_l) * 4u + kwR + 64u * i); }, rsXTfm, 32,
                    nullptr, 0u, 0x100u + 7, -1, [&](int, int) { return PreE{}; },
                    [&](const float (&v)[MTE], const PreE&, int, int n) {
#pragma unroll
                        for (int m = 0; m < MTE; ++m) tilesE_lds[(n * MTE + m) * L::per_tileE + 64 + tid] = v[m];
                    }))
                return;
            // every row had stopped at the previous step: the launch ends here (each workgroup reads the same final count behind seam 2,
            // which every workgroup has waited for in front of phase D; the half step run so far wrote only states nobody reads)
            if (t > 0 && __hip_atomic_load(a.n_unfinished + a.t_base + t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        }
        // =================== phase E: language-LSTM gates + cell ===================
        if (!run_mt<NBA, MTE, PreE>(
                c, TG_E, [&](int m, int i) { return bload4<0>(rsWil, (unsigned)((gate_row_l + 4 * (tE * MTE + m)) * ldw_l + a.We) * 4u + kwA + 64u * i); }, rsGAfm, KBA,
                a.sync + 3 * SHARDS * LINE, tgt, 0x100u + 3, 13,
                [&](int rb, int) {                               // word-side gates and the h_lang block of phase A (handed off at seam 0)
                    PreE p;
                    const int gr = min(rb * 16 + row, B - 1);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        p.gx[m] = (m < MTE && !GREEDY) ? a.GXL[((long long)t * B + gr) * 4 * R + gate * R + 4 * (tE * MTE + m) + ucol] : 0.f;
                        p.lh[m] = m < MTE ? load_sc1(a.LHP + (((long long)rb * TILES_U + tE * MTE + m) << 8) + tid) : 0.f;
                    }
                    return p;
                },
                [&](const float (&v)[MTE], const PreE& p, int rb, int n) {
                    const int grow = rb * 16 + row;
#pragma unroll
                    for (int m = 0; m < MTE; ++m) {
                        const int tau = tE * MTE + m, unit = 4 * tau + ucol;
                        const float gxw = GREEDY ? tilesE_lds[(n * MTE + m) * L::per_tileE + 64 + tid] + bIh[m] : p.gx[m];
                        const float pre = gxw + (p.lh[m] + v[m]) + bE[m];
                        const float act = gate == 2 ? tanhf(pre) : sigm(pre);
                        if (grow < B) a.SL[((long long)ts * B + grow) * 4 * R + gate * R + unit] = act;
                        const float fg = __shfl_down(act, 4, 64), gg = __shfl_down(act, 8, 64), og = __shfl_down(act, 12, 64);
                        float hn = 0.f;
                        if (gate == 0) {
                            float* cs = tilesE_lds + (n * MTE + m) * L::per_tileE + row * 4 + ucol;
                            const float cn = fg * cs[0] + act * gg;
                            hn = og * tanhf(cn);
                            cs[0] = cn;
                            if (grow < B) {
                                a.CL[(long long)(t + 1) * BR + (long long)grow * R + unit] = cn;
                                HLn[(long long)grow * R + unit] = hn;
                            }
                        }
                        const f32x4 h4 = gather4(hn);
                        if (col == 0) bstore4_sc1(rsHLfm, fm_chunk(rb, tau >> 2, KBR) + (unsigned)(((tau & 3) * 16 + row) * 16), h4);
                    }
                }))
            return;
        stamp(c, 14);
        seam_arrive(a.sync + 4 * SHARDS * LINE, shard);
        stamp(c, 15);
    }
    if (GREEDY) {                                                // the last step's token
        if (!logits_station(a.sync + 4 * SHARDS * LINE, (unsigned)G * (unsigned)a.T)) return;
        seam_arrive(a.sync + 6 * SHARDS * LINE, shard);
        if (!seam_wait(a.sync + 6 * SHARDS * LINE, (unsigned)G, status, c.ok_lds, 0x100u + 6)) return;
        if (!rows_station(a.T - 1)) return;
    }
}

constexpr int PR = 512, PA_ = 1024, PP = 16, PG1 = 2 * PR + PA_;

bool persistent_shape_ok(const ekaid_speaker_seq_t* d) {
    return d->R == PR && d->A == PA_ && d->P == PP && d->G1 == PG1 && d->T >= 1 && d->T <= PMAX_T && d->B >= 1 && (d->We % 4) == 0;
}

struct PersistPlan {
    int G, nrb, mtD, mtE, jobsA, jobsD, jobsE, wp_lds, cus;
    size_t lds_bytes;
    // workspace offsets (bytes)
    size_t o_hl, o_ha, o_g1, o_ga, o_mwpw, o_lhp, o_fm, o_wp, o_cp, o_gemm, o_trace, total;
    ekaid_gemm_t gF[3], gWp, gCp;
    size_t gemm_bytes;
};

bool persistent_plan(const ekaid_speaker_seq_t* d, void* workspace, PersistPlan& pl) {
    if (!persistent_shape_ok(d)) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // one workgroup per CU, and a multiple of the 128 gate tiles so that a workgroup keeps ONE column tile for all its row blocks
    const int G = cus >= 256 ? 256 : (cus >= 128 ? 128 : 0);
    if (!G) return false;
    pl.G = G;
    pl.nrb = (d->B + 15) / 16;
    // column tiles per job of the long contractions: activations are read once per job, so wide jobs at large batch (the fabric
    // volume bounds the step), narrow ones while there are fewer jobs than workgroups
    static const int force_mt = [] { const char* e = getenv("EKAID_PERSIST_MT"); return e ? atoi(e) : 0; }();
    // (bounded by the registers the resident weight fragments take: MT x NB blocks of 4 registers; 32 blocks = 128 registers)
    pl.mtD = 1;
    pl.mtE = pl.nrb >= 8 ? 2 : 1;
    if (force_mt == 1) pl.mtD = pl.mtE = 1;
    auto jobs = [&](int tg) { return (tg * pl.nrb + G - 1) / G; };
    pl.jobsA = jobs(PR / 4);
    pl.jobsD = jobs((PA_ / 16) / pl.mtD);
    pl.jobsE = jobs((PR / 4) / pl.mtE);
    if (pl.jobsA > MAXJ || pl.jobsD > MAXJ || pl.jobsE > MAXJ || (d->B + G - 1) / G > 4) return false;
    using L = Lds<PR, PA_, PP>;
    pl.lds_bytes = (size_t)(L::fixed + pl.jobsA * L::per_jobA + pl.jobsD * pl.mtD * L::per_tileD + pl.jobsE * pl.mtE * L::per_tileE) * sizeof(float);
    if (pl.lds_bytes > 160 * 1024) return false;
    pl.wp_lds = pl.lds_bytes + (size_t)PP * (PR + 4) * sizeof(float) <= 160 * 1024;      // weight_pos next to the job slots when it fits
    if (pl.wp_lds) pl.lds_bytes += (size_t)PP * (PR + 4) * sizeof(float);
    {   // every workgroup resident at once? (the teacher-forced instantiation of this tile grouping; the greedy one is checked by its caller)
        const void* kern = pl.mtE == 2 ? reinterpret_cast<const void*>(spk_seq_fwd_persistent_kernel<PR, PA_, PP, 1, 2, false>)
                                       : reinterpret_cast<const void*>(spk_seq_fwd_persistent_kernel<PR, PA_, PP, 1, 1, false>);
        if (!grid_fits(kern, G, cus, pl.lds_bytes)) return false;
        pl.cus = cus;
    }
    const size_t Bp = 16 * (size_t)pl.nrb, f = sizeof(float);
    size_t o = al256((size_t)SYNC_WORDS * 4);
    pl.o_hl = o; o += al256(Bp * PR * f);
    pl.o_ha = o; o += al256(2 * Bp * PR * f);
    pl.o_g1 = o; o += al256(Bp * PG1 * f);
    pl.o_ga = o; o += al256(Bp * PA_ * f);
    pl.o_mwpw = o; o += al256(Bp * 32 * f);
    pl.o_lhp = o; o += al256(Bp * 4 * PR * f);
    pl.o_fm = o; o += al256(3 * (size_t)d->B * PG1 * f);
    pl.o_wp = o; o += al256((size_t)PG1 * PP * f);
    pl.o_cp = o; o += al256((size_t)PG1 * f);
    char* ws = (char*)workspace;
    float* Fm = ws ? (float*)(ws + pl.o_fm) : nullptr;
    const float* feats[3] = {d->bef, d->diff, d->aft};
    static const float dummy = 0.f;
    pl.gemm_bytes = 0;
    for (int m = 0; m < 3; ++m) {       // F_m = feat_m . W_g1[:, 2R:]^T
        pl.gF[m] = dense_desc(feats[m] ? feats[m] : &dummy, PA_, d->w_g1 ? d->w_g1 + 2 * PR : &dummy, PG1, 1, Fm ? Fm + (size_t)m * d->B * PG1 : nullptr, PG1, d->B, PG1, PA_);
        pl.gemm_bytes = std::max(pl.gemm_bytes, ekaid_gemm_workspace_bytes(&pl.gF[m]));
    }
    // Wp = W_g1[:, R:2R] . w_p2 (G1 x P), cp = W_g1[:, R:2R] . b_p2 (G1)
    pl.gWp = dense_desc(d->w_g1 ? d->w_g1 + PR : &dummy, PG1, d->w_p2 ? d->w_p2 : &dummy, PP, 0, ws ? (float*)(ws + pl.o_wp) : nullptr, PP, PG1, PP, PR);
    pl.gCp = dense_desc(d->w_g1 ? d->w_g1 + PR : &dummy, PG1, d->b_p2 ? d->b_p2 : &dummy, 1, 0, ws ? (float*)(ws + pl.o_cp) : nullptr, 1, PG1, 1, PR);
    pl.gemm_bytes = std::max(pl.gemm_bytes, std::max(ekaid_gemm_workspace_bytes(&pl.gWp), ekaid_gemm_workspace_bytes(&pl.gCp)));
    pl.o_gemm = o; o += al256(pl.gemm_bytes);
    pl.o_trace = o;
    if (persist_trace_enabled()) o += al256((size_t)G * d->T * 16 * sizeof(unsigned long long));
    pl.total = o;
    return true;
}

template <int MTD, int MTE, bool GREEDY = false>
int launch_fwd(const PFwd& a, const PersistPlan& pl, hipStream_t st) {
    auto kern = spk_seq_fwd_persistent_kernel<PR, PA_, PP, MTD, MTE, GREEDY>;
    if (func_attr_once(reinterpret_cast<const void*>(kern))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("ekaid_speaker_seq_fwd_persistent: cannot raise the dynamic LDS limit");
            return EKAID_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(pl.G), dim3(256), pl.lds_bytes, st, a);
    count_launch(GREEDY ? EKAID_COUNT_GREEDY_PERSISTENT : EKAID_COUNT_SPK_FWD_PERSISTENT);
    return check_launch("speaker_seq_fwd_persistent");
}

}  // namespace
}  // namespace ekaid

using namespace ekaid;

extern "C" size_t ekaid_speaker_seq_persistent_workspace_bytes(const ekaid_speaker_seq_t* d) {
    PersistPlan pl;
    if (!d || !persistent_plan(d, nullptr, pl)) return 0;
    return pl.total;
}

namespace ekaid {
namespace {
// builds the kernel arguments, runs the prologue products and launches; `gq` != NULL: greedy decoding of steps [t0, t0 + d->T)
int launch_persistent_fwd(const ekaid_speaker_seq_t* d, const ekaid_speaker_greedy_t* gq, int t0, void* workspace, size_t workspace_bytes, void* stream,
                          const char* who) {
    PersistPlan pl;
    if (!persistent_plan(d, workspace, pl)) {
        set_error("%s: unsupported shape (R=%d A=%d P=%d G1=%d T=%d B=%d) or device", who, d->R, d->A, d->P, d->G1, d->T, d->B);
        return EKAID_ERR_UNSUPPORTED;
    }
    const size_t Bp = 16 * (size_t)pl.nrb;
    const size_t o_xt = pl.total, o_log = o_xt + al256(Bp * 512 * sizeof(float)), total = gq ? o_log + al256(Bp * 160 * sizeof(float)) : pl.total;
    if (!workspace || workspace_bytes < total || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        set_error("%s: workspace too small or misaligned (%zu < %zu)", who, workspace_bytes, total);
        return EKAID_ERR_ARG;
    }
    if (!d->seeds && (d->p_lm > 0.f || d->p_pos > 0.f)) { set_error("%s: dropout without seeds", who); return EKAID_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PFwd a;
    memset(&a, 0, sizeof(a));
    a.T = d->T; a.B = d->B; a.G = pl.G; a.nrb = pl.nrb; a.mtD = pl.mtD; a.mtE = pl.mtE;
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
    a.status = seq_status_ptr();
    if (!a.status) { set_error("%s: cannot allocate the status word", who); return EKAID_ERR_LAUNCH; }
    a.trace = persist_trace_enabled() ? reinterpret_cast<unsigned long long*>(ws + pl.o_trace) : nullptr;
    a.HLfm = (float*)(ws + pl.o_hl); a.HAfm = (float*)(ws + pl.o_ha); a.G1fm = (float*)(ws + pl.o_g1); a.GAfm = (float*)(ws + pl.o_ga);
    a.MWPW = (float*)(ws + pl.o_mwpw); a.LHP = (float*)(ws + pl.o_lhp);
    a.Fm = (const float*)(ws + pl.o_fm); a.Wp = (const float*)(ws + pl.o_wp); a.cp = (const float*)(ws + pl.o_cp);
    a.We = d->We; a.jobsA = pl.jobsA; a.jobsD = pl.jobsD; a.jobsE = pl.jobsE; a.wp_lds = pl.wp_lds;
    if (d->seeds) memcpy(a.seeds, d->seeds, sizeof(uint64_t) * 3 * d->T);
    if (gq) {
        a.V = gq->V; a.seq_len = gq->seq_len; a.t_base = t0;
        a.emb = gq->emb; a.bih_l = gq->bih_l; a.w_lo = gq->w_lo; a.b_lo = gq->b_lo;
        a.it = (long long*)gq->it; a.unfinished = gq->unfinished; a.seq = (long long*)gq->seq; a.seq_logp = gq->seq_logp; a.n_unfinished = gq->n_unfinished;
        a.XTfm = (float*)(ws + o_xt); a.LOG = (float*)(ws + o_log);
    }
    const void* ptrs[] = {a.Wcat, a.whh_a, a.bhh_a, a.w_fc, a.b_fc, a.b_p1, a.w_wp, a.b_wp, a.w_p2, a.b_p2, a.w_g1, a.b_g1, a.w_g2, a.b_g2, a.wih_l,
                          a.bhh_l, a.bef, a.diff, a.aft, a.GXA, gq ? (const void*)a.GXA : (const void*)a.GXL, a.HA, a.CA, a.HL, a.CL, a.SA, a.SL, a.G1D, a.VP,
                          a.DP, a.PW, a.PA, a.AT, a.GT, a.GA};
    for (const void* p : ptrs)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) { set_error("%s: null or misaligned (16 B) pointer in the descriptor", who); return EKAID_ERR_ARG; }
    if (!a.MW || (reinterpret_cast<uintptr_t>(a.MW) & 3)) { set_error("%s: null or misaligned module-weight buffer", who); return EKAID_ERR_ARG; }
    // the linear maps of gate-MLP layer 1 (5 small products in front of the sequence kernel)
    void* gws = ws + pl.o_gemm;
    for (int m = 0; m < 3; ++m) { const int rc = ekaid_gemm_f32(&pl.gF[m], gws, pl.gemm_bytes, stream); if (rc) return rc; }
    { const int rc = ekaid_gemm_f32(&pl.gWp, gws, pl.gemm_bytes, stream); if (rc) return rc; }
    { const int rc = ekaid_gemm_f32(&pl.gCp, gws, pl.gemm_bytes, stream); if (rc) return rc; }
    // seam counters start at zero in EVERY launch (a memset node when captured)
    if (hipMemsetAsync(workspace, 0, al256((size_t)SYNC_WORDS * 4), st) != hipSuccess) { (void)hipGetLastError(); set_error("%s: memset failed", who); return EKAID_ERR_LAUNCH; }
    if (gq) {
        if (pl.mtD == 1 && pl.mtE == 1) return launch_fwd<1, 1, true>(a, pl, st);
        if (pl.mtD == 1 && pl.mtE == 2) return launch_fwd<1, 2, true>(a, pl, st);
    } else {
        if (pl.mtD == 1 && pl.mtE == 1) return launch_fwd<1, 1>(a, pl, st);
        if (pl.mtD == 1 && pl.mtE == 2) return launch_fwd<1, 2>(a, pl, st);
    }
    set_error("%s: no kernel for tile groups %d / %d", who, pl.mtD, pl.mtE);
    return EKAID_ERR_UNSUPPORTED;
}
}  // namespace
}  // namespace ekaid

extern "C" int ekaid_speaker_seq_fwd_persistent(const ekaid_speaker_seq_t* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d) { set_error("ekaid_speaker_seq_fwd_persistent: null descriptor"); return EKAID_ERR_ARG; }
    return launch_persistent_fwd(d, nullptr, 0, workspace, workspace_bytes, stream, "ekaid_speaker_seq_fwd_persistent");
}

/* Greedy decoding of steps [t0, t1) in ONE persistent launch: the forward kernel with three more stations per step -- the word-side
 * gates of the language LSTM from the fed-back tokens, the logits, and a row-local station (log-softmax, arg-max with lowest-index
 * ties, token 0 forbidden at step 0, stop flags, seq / seq_logprobs, next token).  Same descriptor and results as
 * ekaid_speaker_greedy_steps (token ids identical unless two log-probs tie within f32 rounding). */
extern "C" size_t ekaid_speaker_greedy_persistent_workspace_bytes(const ekaid_speaker_greedy_t* q, int steps) {
    if (!q || steps < 1 || q->V < 1 || q->V > 160 || q->f.We > 512) return 0;
    ekaid_speaker_seq_t d = q->f;
    d.T = steps;
    PersistPlan pl;
    if (!persistent_plan(&d, nullptr, pl)) return 0;
    {
        const void* kern = pl.mtE == 2 ? reinterpret_cast<const void*>(spk_seq_fwd_persistent_kernel<PR, PA_, PP, 1, 2, true>)
                                       : reinterpret_cast<const void*>(spk_seq_fwd_persistent_kernel<PR, PA_, PP, 1, 1, true>);
        if (!grid_fits(kern, pl.G, pl.cus, pl.lds_bytes)) return 0;
    }
    const size_t Bp = 16 * (size_t)pl.nrb;
    return pl.total + al256(Bp * 512 * sizeof(float)) + al256(Bp * 160 * sizeof(float));
}

extern "C" int ekaid_speaker_greedy_steps_persistent(const ekaid_speaker_greedy_t* q, int t0, int t1, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
    if (!q || t0 < 0 || t1 <= t0 || t1 > q->f.T || q->seq_len < t1 || !q->emb || !q->bih_l || !q->w_lo || !q->b_lo || !q->it || !q->unfinished ||
        !q->seq || !q->seq_logp || !q->n_unfinished || q->V < 1 || q->V > 160 || q->f.We > 512) {
        set_error("ekaid_speaker_greedy_steps_persistent: bad arguments");
        return EKAID_ERR_ARG;
    }
    if (q->f.p_lm != 0.f || q->f.p_pos != 0.f) { set_error("ekaid_speaker_greedy_steps_persistent: eval mode only"); return EKAID_ERR_UNSUPPORTED; }
    ekaid_speaker_seq_t d = q->f;
    const size_t BR = (size_t)d.B * d.R;
    d.T = t1 - t0;
    d.seeds = nullptr;
    d.HA = q->f.HA + t0 * BR; d.CA = q->f.CA + t0 * BR; d.HL = q->f.HL + t0 * BR; d.CL = q->f.CL + t0 * BR;
    d.MW = q->f.MW + (size_t)t0 * d.B * 3;
    return launch_persistent_fwd(&d, q, t0, workspace, workspace_bytes, stream, "ekaid_speaker_greedy_steps_persistent");
}

/* device pointer to the per-device status word of the persistent sequence kernels: 0 = every launch so far was fine; 0x100 / 0x200 /
 * 0x300 + seam = a seam of a decoder forward / backward / GRU launch did not open in time and that launch (and, the word being sticky,
 * every later one) was abandoned.  (The argument is ignored: kept for callers of ABI 6 drafts.) */
extern "C" const unsigned* ekaid_speaker_seq_persistent_status_ptr(const void* workspace) {
    (void)workspace;
    return seq_status_ptr();
}
