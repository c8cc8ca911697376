// Persistent backward of the teacher-forced answer decoder: back-propagation through all T steps in ONE launch.
//
// ref: the autograd of dynamic_speaker_change_pos.py:94-131 inside the loop :196-222 -- here explicit BPTT over the saved tensors of
// ekaid_speaker_seq_fwd, the same contract as the step-wise ekaid_speaker_seq_bwd (csrc/sequence.hip: 8 launches per step).
//
// Same machinery as the forward (csrc/persist.hip, csrc/persist.h): one workgroup per CU resident for the whole sequence, 16 x 16
// f32-MFMA jobs with the full contraction, fragment-major hand-offs behind device-scope seams, weights (here: transposed copies made
// once per call, so that every contraction reads its weight rows k-contiguously) resident per phase.  A step t = T-1 .. 0:
//   P1  language-LSTM cell backward, element-wise, by the owner of (16 hidden units, 16 rows): upstream = the carry it accumulated
//       itself during step t+1 + the output-head gradient                                               -> d lang gates   [seam 1]
//   P2  d gated = d lang gates . W_ih_lang[:, We:]   (K = 4R; 64 tiles): gate backward -> d z2; per-row partial sums of
//       <d gated * gate, feat_m> (the module-mix backward needs the row totals)                                           [seam 2]
//   P3  d g1d = d z2 . W_g2   (K = A; 128 tiles): dropout / relu backward -> d gate1x pre-activation; the product
//       d [pos_feat | att] = d g1pre . W_g1[:, R:] of the step-wise path (K = G1, 1536 columns, the largest of the chain) is NOT
//       formed: what it feeds is linear, so per-row partial sums of <d g1pre, F_m> and d g1pre . Wp (F_m = feat_m W_att^T,
//       Wp = W_pp W_p2, as in the forward) give the module-weight and POS-weight gradients, and d feat_m = (sum_t mw_m d g1pre) W_att
//       is accumulated per tile (dFm) and finished by one product per feature after the sequence                          [seam 3]
//   P4  row-local: totals of the partials -> softmax backward of the module weights (d z_fc) and of the POS weights (-> d dpos),
//       d vpos -> d pos1 pre-activation                                                                                   [seam 4]
//   P5  attention-LSTM cell backward, element-wise, by the same owners as P1 (upstream = own carry + d z_fc . w_fc)
//                                                                                                        -> d att gates   [seam 5]
//   P6  carries for step t-1, by the owners of the hidden units: d h_att = d att gates . W_hh_att, d h_lang = DCAT_t . Wcat -- the
//       latter as four partial products over the column blocks of DCAT (language gates, gate1x, pos1, attention gates), each run as
//       soon as its block exists (the first two in the shadow of seams 2 and 4), summed in LDS.  No seam follows: P1 of step t-1 is
//       local to the same workgroup.
// Outputs as ekaid_speaker_seq_bwd (DCAT, DZFC, DWP, DZ2, d_bef/d_diff/d_aft += the gated-path part) except DPP, which the caller
// forms as one time-batched product DCAT[:, gate1x block] . W_g1[:, R:2R], plus dFm [3][B][G1] (see P3).
#include "../../include/ekaid_hip.h"
#include "persist.h"

namespace ekaid {
namespace {

struct PBwd {
    int T, B, G, nrb;
    float p_lm, p_pos, inv_lm, inv_pos;
    const unsigned long long* ep;
    const float *WT1 /* [A][4R] */, *WT2 /* [G1][A] */, *WT3 /* [R][4R] */, *WT4 /* [R][NW] */;
    const float *w_fc, *w_wp, *bef, *diff, *aft, *Fm, *Wp;
    const float *CA, *CL, *SA, *SL, *G1D, *MW, *VP, *PW, *AT, *GT;
    const float *dHLo, *dDPl;
    float *DCAT, *DZFC, *DWP, *DZ2, *d_bef, *d_diff, *d_aft, *dFm;
    float *DGLfm, *DZ2fm, *DG1fm, *DP1fm, *DGAfm, *PART1 /* [nrb][G1/16][16][20] */, *PARTG /* [nrb][A/16][16][4] */, *DZ /* [Bp][4] */;
    unsigned* sync;
    unsigned* status;                 // per-device sticky word: non-zero = a seam timed out (api.hip: seq_status_ptr)
    unsigned long long* trace;
    int jobsU, jobs2, jobs3;
    uint64_t seeds[3 * PMAX_T];
};

struct PreP2 { float att, gate, w0, w1, w2; };
struct PreP3 { float g1d, w0, w1, w2; };
struct PreNone { float x; };

// sum over the 16 lanes of a tile row (lanes of equal tid >> 4): every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ void lstm_cell_bwd(float ig, float fg, float gg, float og, float c_prev, float c_cur, float dh, float dc_in,
                                              float (&dg)[4], float& dc_out) {
    const float tc = tanhf(c_cur);
    const float dct = dc_in + dh * og * (1.f - tc * tc);
    dg[0] = dct * gg * ig * (1.f - ig);
    dg[1] = dct * c_prev * fg * (1.f - fg);
    dg[2] = dct * ig * (1.f - gg * gg);
    dg[3] = dh * tc * og * (1.f - og);
    dc_out = dct * fg;
}

template <int R, int A, int P>
__global__ __launch_bounds__(256, 1) void spk_seq_bwd_persistent_kernel(const PBwd a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int G1 = 2 * R + A, NW = 9 * R + G1, oP1 = 4 * R, oG1 = 5 * R, oGL = 5 * R + G1;
    constexpr int NB4R = 4 * R / 64, NBA = A / 64, NBG = G1 / 64, NBR = R / 64;
    constexpr int KB4R = 4 * R / 16, KBA = A / 16, KBG = G1 / 16, KBR = R / 16;
    constexpr int TU = R / 16, T2 = A / 16, T3 = G1 / 16;
    static_assert(NB4R == 32 && NBG == 32, "the long contractions are 2048 deep");
    // LDS carve (floats)
    constexpr int L_flag = 0, L_red = 4, L_rs = L_red + 2 * 5120 /* 96: wave sums of the row phase */, L_tot = L_rs + 96 /* 32 */, L_gp = L_tot + 32 /* 16 */,
                  L_fixed = L_gp + 16 + 12;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int B = a.B, G = a.G, nrb = a.nrb;
    const int l15 = lane & 15, kq4 = (lane >> 4) * 4;
    const int row = tid >> 4, col = tid & 15;
    const unsigned shard = xcc_id();
    unsigned* const status = a.status;
    float* const slotsU = lds + L_fixed;                         // per job: dhl_acc | dha_acc | dc_l | dc_a   (4 x 256)
    float* const slots2 = slotsU + a.jobsU * 1024;               // per job: feat_m slices (3 x 256) | d feat_m accumulators (3 x 256)
    float* const slots3 = slots2 + a.jobs2 * 1536;               // per job: F_m slices (3 x 256) | dF_m accumulators (3 x 256)
    Ctx c = {tid, lane, wave, wg, G, nrb, B, row, col, lds + L_red, 0, status, reinterpret_cast<int*>(lds + L_flag), nullptr};
    const long long BR = (long long)B * R, Bp = 16ll * nrb;

    // ---- ownership
    const int tUo = wg % TU, rbU0 = wg / TU, drbU = G / TU;
    const int njU = wg < TU * nrb ? (TU * nrb - wg + G - 1) / G : 0;
    const int t2 = wg % T2, rb20 = wg / T2, drb2 = G / T2;
    const int nj2 = wg < T2 * nrb ? (T2 * nrb - wg + G - 1) / G : 0;
    const int t3 = wg % T3, rb30 = wg / T3, drb3 = G / T3;
    const int nj3 = wg < T3 * nrb ? (T3 * nrb - wg + G - 1) / G : 0;
    const int unitU = 16 * tUo + col, n2 = 16 * t2 + col, n3 = 16 * t3 + col;

    // ---- one-time preloads
    for (int n = 0; n < njU; ++n)
        for (int i = tid; i < 1024; i += 256) slotsU[n * 1024 + i] = 0.f;
    for (int n = 0; n < nj2; ++n) {
        const int gr = min((rb20 + n * drb2) * 16 + row, B - 1);
        float* s = slots2 + n * 1536;
        const long long o = (long long)gr * A + n2;
        s[tid] = a.bef[o]; s[256 + tid] = a.diff[o]; s[512 + tid] = a.aft[o];
        s[768 + tid] = 0.f; s[1024 + tid] = 0.f; s[1280 + tid] = 0.f;
    }
    for (int n = 0; n < nj3; ++n) {
        const int gr = min((rb30 + n * drb3) * 16 + row, B - 1);
        float* s = slots3 + n * 1536;
#pragma unroll
        for (int m = 0; m < 3; ++m) { s[m * 256 + tid] = a.Fm[((long long)m * B + gr) * G1 + n3]; s[768 + m * 256 + tid] = 0.f; }
    }
    float wp[P];
#pragma unroll
    for (int j = 0; j < P; ++j) wp[j] = a.Wp[(long long)n3 * P + j];
    const float wfc0 = a.w_fc[unitU], wfc1 = a.w_fc[R + unitU], wfc2 = a.w_fc[2 * R + unitU];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsWT1 = make_rsrc(a.WT1, 4ll * A * 4 * R), rsWT2 = make_rsrc(a.WT2, 4ll * G1 * A), rsWT3 = make_rsrc(a.WT3, 4ll * R * 4 * R),
                                 rsWT4 = make_rsrc(a.WT4, 4ll * R * NW);
    const __amdgpu_buffer_rsrc_t rsDGL = make_rsrc(a.DGLfm, Bp * 4 * R * 4), rsDZ2 = make_rsrc(a.DZ2fm, Bp * A * 4), rsDG1 = make_rsrc(a.DG1fm, Bp * G1 * 4),
                                 rsDP1 = make_rsrc(a.DP1fm, Bp * R * 4), rsDGA = make_rsrc(a.DGAfm, Bp * 4 * R * 4);
    const unsigned kw4R = (unsigned)(wave * R + kq4) * 4u;               // this lane's first k of a 4R-long contraction (wave quarter = R)
    const unsigned kwA = (unsigned)(wave * (A / 4) + kq4) * 4u;
    const unsigned kwG = (unsigned)(wave * (G1 / 4) + kq4) * 4u;
    const unsigned kwR = (unsigned)(wave * (R / 4) + kq4) * 4u;
    auto nopre = [](int, int) { return PreNone{0.f}; };
    // carry accumulation into slot column `which` (0: d h_lang, 1: d h_att): add or set
    auto acc_epi = [&](int which, bool set) {
        return [=](const float (&v)[1], const PreNone&, int, int n) {
            float* p = slotsU + n * 1024 + which * 256 + tid;
            *p = set ? v[0] : *p + v[0];
        };
    };

    for (int s = 0; s < a.T; ++s) {
        const int t = a.T - 1 - s;
        const DropSeed sv = {a.seeds[3 * t], a.ep}, sd = {a.seeds[3 * t + 1], a.ep}, sg = {a.seeds[3 * t + 2], a.ep};
        const unsigned tgt = (unsigned)G * (unsigned)(s + 1);
        float* DCATt = a.DCAT + (long long)t * B * NW;
        c.trace = a.trace ? a.trace + ((long long)wg * a.T + s) * 16 : nullptr;
        stamp(c, 0);

        // =================== P1: language-LSTM cell backward (owners of 16 units x 16 rows) ===================
        for (int n = 0; n < njU; ++n) {
            const int rb = rbU0 + n * drbU, grow = rb * 16 + row, gr = min(grow, B - 1);
            float* sl = slotsU + n * 1024;
            const float* sp = a.SL + ((long long)t * B + gr) * 4 * R;
            const float dh = sl[tid] + a.dHLo[((long long)t * B + gr) * R + unitU];
            float dg[4], dc;
            lstm_cell_bwd(sp[unitU], sp[R + unitU], sp[2 * R + unitU], sp[3 * R + unitU], a.CL[(long long)t * BR + (long long)gr * R + unitU],
                          a.CL[(long long)(t + 1) * BR + (long long)gr * R + unitU], dh, sl[512 + tid], dg, dc);
            sl[512 + tid] = dc;
            sl[tid] = 0.f;                                       // the carry of step t-1 is accumulated from here on
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (grow < B) DCATt[(long long)grow * NW + oGL + g * R + unitU] = dg[g];
                const f32x4 q = gather4(dg[g]);
                if ((col & 3) == 0) bstore4_sc1(rsDGL, fm_chunk(rb, g * TU + tUo, KB4R) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
            }
        }
        stamp(c, 1);
        seam_arrive(a.sync + 0 * SHARDS * LINE, shard);
        stamp(c, 2);
        // ---- shadow of seam 1: the attention-LSTM carry of the PREVIOUS step, d h_att_t = d att gates_{t+1} . W_hh_att -- not needed before
        // P5, so it leaves the critical stretch between seam 5 and P1 (the hand-off buffer is rewritten only in P5 of this step)
        if (s > 0)
            run_mt<NB4R, 1, PreNone>(c, TU, [&](int, int i) { return bload4<0>(rsWT3, (unsigned)((16 * tUo + l15) * 4 * R) * 4u + kw4R + 64u * i); }, rsDGA,
                                     KB4R, nullptr, 0u, 0u, -1, nopre, acc_epi(1, true));
        // =================== P2: d gated, gate backward, gated-path partials of the module mix ===================
        if (!run_mt<NB4R, 1, PreP2>(
                c, T2, [&](int, int i) { return bload4<0>(rsWT1, (unsigned)((16 * t2 + l15) * 4 * R) * 4u + kw4R + 64u * i); }, rsDGL, KB4R,
                a.sync + 0 * SHARDS * LINE, tgt, 0x200u + 0, 3,
                [&](int rb, int) {
                    const int gr = min(rb * 16 + row, B - 1);
                    const float* mw = a.MW + ((long long)t * B + gr) * 3;
                    const long long o = ((long long)t * B + gr) * A + n2;
                    return PreP2{a.AT[o], a.GT[o], mw[0], mw[1], mw[2]};
                },
                [&](const float (&v)[1], const PreP2& p, int rb, int n) {
                    const int grow = rb * 16 + row;
                    const float dgd = v[0];
                    const float dz2 = dgd * p.att * p.gate * (1.f - p.gate);
                    if (grow < B) a.DZ2[((long long)t * B + grow) * A + n2] = dz2;
                    const f32x4 q = gather4(dz2);
                    if ((col & 3) == 0) bstore4_sc1(rsDZ2, fm_chunk(rb, t2, KBA) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
                    const float dgg = dgd * p.gate;              // gradient reaching att through the gating product
                    float* sl = slots2 + n * 1536;
                    sl[768 + tid] += p.w0 * dgg; sl[1024 + tid] += p.w1 * dgg; sl[1280 + tid] += p.w2 * dgg;
                    const float p0 = row16_sum(dgg * sl[tid]), p1 = row16_sum(dgg * sl[256 + tid]), p2 = row16_sum(dgg * sl[512 + tid]);
                    if (col < 3) store_sc1(a.PARTG + (((long long)rb * T2 + t2) * 16 + row) * 4 + col, col == 0 ? p0 : (col == 1 ? p1 : p2));
                }))
            return;
        stamp(c, 4);
        seam_arrive(a.sync + 1 * SHARDS * LINE, shard);
        stamp(c, 5);
        // ---- shadow of seam 2: d h_lang_{t-1} += d lang gates . Wcat[lang-gate block]
        if (t > 0)
            run_mt<NB4R, 1, PreNone>(c, TU, [&](int, int i) { return bload4<0>(rsWT4, (unsigned)((16 * tUo + l15) * NW + oGL) * 4u + kw4R + 64u * i); }, rsDGL,
                                     KB4R, nullptr, 0u, 0u, -1, nopre, acc_epi(0, false));
        stamp(c, 6);
        // =================== P3: d g1d, dropout / relu backward, partials of the module-weight and POS-weight gradients ===================
        if (!run_mt<NBA, 1, PreP3>(
                c, T3, [&](int, int i) { return bload4<0>(rsWT2, (unsigned)((16 * t3 + l15) * A) * 4u + kwA + 64u * i); }, rsDZ2, KBA,
                a.sync + 1 * SHARDS * LINE, tgt, 0x200u + 1, 7,
                [&](int rb, int) {
                    const int gr = min(rb * 16 + row, B - 1);
                    const float* mw = a.MW + ((long long)t * B + gr) * 3;
                    return PreP3{a.G1D[((long long)t * B + gr) * G1 + n3], mw[0], mw[1], mw[2]};
                },
                [&](const float (&v)[1], const PreP3& p, int rb, int n) {
                    const int grow = rb * 16 + row;
                    const float dg1 = v[0] * dropout_scale(sg, (uint64_t)grow * G1 + n3, a.p_lm, a.inv_lm) * (p.g1d > 0.f ? 1.f : 0.f);
                    if (grow < B) DCATt[(long long)grow * NW + oG1 + n3] = dg1;
                    const f32x4 q = gather4(dg1);
                    if ((col & 3) == 0) bstore4_sc1(rsDG1, fm_chunk(rb, t3, KBG) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
                    float* sl = slots3 + n * 1536;
                    sl[768 + tid] += p.w0 * dg1; sl[1024 + tid] += p.w1 * dg1; sl[1280 + tid] += p.w2 * dg1;
                    // per-row partial sums over this tile's 16 columns: 3 x <d g1pre, F_m> and 16 x d g1pre . Wp[:, j]
                    float tot[20];
#pragma unroll
                    for (int m = 0; m < 3; ++m) tot[m] = row16_sum(dg1 * sl[m * 256 + tid]);
#pragma unroll
                    for (int j = 0; j < P; ++j) tot[3 + j] = row16_sum(dg1 * wp[j]);
                    tot[19] = 0.f;
                    if (col < 5) {                               // lane col stores floats [4 col, 4 col + 4) of the row's 20-float record
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = tot[e];
#pragma unroll
                            for (int cc = 1; cc < 5; ++cc) x = col == cc ? tot[4 * cc + e] : x;
                            o[e] = x;
                        }
                        float* dst = a.PART1 + (((long long)rb * T3 + t3) * 16 + row) * 20 + 4 * col;
#pragma unroll
                        for (int e = 0; e < 4; ++e) store_sc1(dst + e, o[e]);
                    }
                }))
            return;
        stamp(c, 8);
        seam_arrive(a.sync + 2 * SHARDS * LINE, shard);
        stamp(c, 9);
        if (!seam_wait(a.sync + 2 * SHARDS * LINE, tgt, status, c.ok_lds, 0x200u + 2)) return;
        stamp(c, 10);
        // =================== P4: row-local -- totals, softmax backwards, d vpos ===================
        for (int r = wg, rs = 0; r < B; r += G, ++rs) {
            if (rs) __syncthreads();
            const int rb = r >> 4, rr = r & 15;
            {   // 128 tiles x 20 floats: thread (tile = tid >> 1, half = tid & 1) loads 10, wave-reduces over the lanes of equal half
                const float* src = a.PART1 + (((long long)rb * T3 + (tid >> 1)) * 16 + rr) * 20 + 10 * (tid & 1);
                float x[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) x[i] = load_sc1(src + i);
#pragma unroll
                for (int i = 0; i < 10; ++i) {
#pragma unroll
                    for (int o = 2; o < 64; o <<= 1) x[i] += __shfl_xor(x[i], o, 64);
                }
                if (lane < 2) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) lds[L_rs + wave * 20 + lane * 10 + i] = x[i];
                }
                float g3 = 0.f;
                if (tid < 64 * 3) {                              // gated-path partials: 64 tiles x 3: wave m sums value m over the tiles
                    g3 = load_sc1(a.PARTG + (((long long)rb * T2 + lane) * 16 + rr) * 4 + wave);
                    g3 = wave_sum(g3);
                    if (lane == 0) lds[L_rs + 80 + wave] = g3;
                }
            }
            __syncthreads();
            if (tid < 20) {
                float x = (lds[L_rs + tid] + lds[L_rs + 20 + tid]) + (lds[L_rs + 40 + tid] + lds[L_rs + 60 + tid]);
                if (tid < 3) x += lds[L_rs + 80 + tid];
                lds[L_tot + tid] = x;
            }
            __syncthreads();
            if (tid < 3) {                                       // module-weight softmax backward
                const float* mw = a.MW + ((long long)t * B + r) * 3;
                const float w0 = mw[0], w1 = mw[1], w2 = mw[2];
                const float s0 = lds[L_tot], s1 = lds[L_tot + 1], s2 = lds[L_tot + 2];
                const float dot = w0 * s0 + w1 * s1 + w2 * s2;
                const float dz = tid == 0 ? w0 * (s0 - dot) : (tid == 1 ? w1 * (s1 - dot) : w2 * (s2 - dot));
                a.DZFC[((long long)t * B + r) * 3 + tid] = dz;
                store_sc1(a.DZ + (long long)r * 4 + tid, dz);
            } else if (tid >= 64 && tid < 64 + P) {              // POS-weight softmax backward (+ the POS head), dropout backward: 16 lanes
                const int j = tid - 64;
                const float pw = a.PW[((long long)t * B + r) * P + j], sgj = lds[L_tot + 3 + j];
                float dt = pw * sgj;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) dt += __shfl_xor(dt, o, 16);
                float gj = pw * (sgj - dt);
                if (a.dDPl) gj += a.dDPl[((long long)t * B + r) * P + j];
                gj *= dropout_scale(sd, (uint64_t)r * P + j, a.p_pos, a.inv_pos);
                lds[L_gp + j] = gj;
                a.DWP[((long long)t * B + r) * P + j] = gj;
            }
            __syncthreads();
            if (tid < R / 4) {                                   // d vpos = w_wp^T . d dpos; dropout / relu backward -> d pos1 pre-activation
                const int k = 4 * tid;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < P; ++j) acc += lds[L_gp + j] * *reinterpret_cast<const f32x4*>(a.w_wp + (long long)j * R + k);
                const f32x4 y = *reinterpret_cast<const f32x4*>(a.VP + ((long long)t * B + r) * R + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] *= dropout_scale(sv, (uint64_t)r * R + k + e, a.p_lm, a.inv_lm) * (y[e] > 0.f ? 1.f : 0.f);
                *reinterpret_cast<f32x4*>(DCATt + (long long)r * NW + oP1 + k) = acc;
                bstore4_sc1(rsDP1, fm_chunk(rb, tid >> 2, KBR) + (unsigned)(((tid & 3) * 16 + rr) * 16), acc);
            }
        }
        stamp(c, 11);
        seam_arrive(a.sync + 3 * SHARDS * LINE, shard);
        stamp(c, 12);
        // ---- shadow of seam 4: d h_lang_{t-1} += d gate1x . Wcat[gate1x block]
        if (t > 0)
            run_mt<NBG, 1, PreNone>(c, TU, [&](int, int i) { return bload4<0>(rsWT4, (unsigned)((16 * tUo + l15) * NW + oG1) * 4u + kwG + 64u * i); }, rsDG1,
                                    KBG, nullptr, 0u, 0u, -1, nopre, acc_epi(0, false));
        stamp(c, 13);
        if (!seam_wait(a.sync + 3 * SHARDS * LINE, tgt, status, c.ok_lds, 0x200u + 3)) return;
        // =================== P5: attention-LSTM cell backward (same owners as P1) ===================
        for (int n = 0; n < njU; ++n) {
            const int rb = rbU0 + n * drbU, grow = rb * 16 + row, gr = min(grow, B - 1);
            float* sl = slotsU + n * 1024;
            const float* sp = a.SA + ((long long)t * B + gr) * 4 * R;
            const float* dz = a.DZ + (long long)gr * 4;
            const float dh = sl[256 + tid] + (load_sc1(dz) * wfc0 + load_sc1(dz + 1) * wfc1 + load_sc1(dz + 2) * wfc2);
            float dg[4], dc;
            lstm_cell_bwd(sp[unitU], sp[R + unitU], sp[2 * R + unitU], sp[3 * R + unitU], a.CA[(long long)t * BR + (long long)gr * R + unitU],
                          a.CA[(long long)(t + 1) * BR + (long long)gr * R + unitU], dh, sl[768 + tid], dg, dc);
            sl[768 + tid] = dc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (grow < B) DCATt[(long long)grow * NW + g * R + unitU] = dg[g];
                const f32x4 q = gather4(dg[g]);
                if ((col & 3) == 0) bstore4_sc1(rsDGA, fm_chunk(rb, g * TU + tUo, KB4R) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
            }
        }
        stamp(c, 14);
        seam_arrive(a.sync + 4 * SHARDS * LINE, shard);
        // =================== P6: the carries of step t-1 ===================
        if (t > 0) {
            if (!run_mt<NB4R, 1, PreNone>(c, TU, [&](int, int i) { return bload4<0>(rsWT4, (unsigned)((16 * tUo + l15) * NW) * 4u + kw4R + 64u * i); }, rsDGA,
                                          KB4R, a.sync + 4 * SHARDS * LINE, tgt, 0x200u + 4, -1, nopre, acc_epi(0, false)))
                return;
            run_mt<NBR, 1, PreNone>(c, TU, [&](int, int i) { return bload4<0>(rsWT4, (unsigned)((16 * tUo + l15) * NW + oP1) * 4u + kwR + 64u * i); }, rsDP1, KBR,
                                    nullptr, 0u, 0u, -1, nopre, acc_epi(0, false));
        }
        stamp(c, 15);
    }

    // ---- the accumulated feature gradients: gated path into d_bef / d_diff / d_aft, gate-MLP path as dFm (finished by the caller)
    for (int n = 0; n < nj2; ++n) {
        const int grow = (rb20 + n * drb2) * 16 + row;
        if (grow < B) {
            const float* sl = slots2 + n * 1536;
            const long long o = (long long)grow * A + n2;
            a.d_bef[o] += sl[768 + tid]; a.d_diff[o] += sl[1024 + tid]; a.d_aft[o] += sl[1280 + tid];
        }
    }
    for (int n = 0; n < nj3; ++n) {
        const int grow = (rb30 + n * drb3) * 16 + row;
        if (grow < B) {
            const float* sl = slots3 + n * 1536;
#pragma unroll
            for (int m = 0; m < 3; ++m) a.dFm[((long long)m * B + grow) * G1 + n3] = sl[768 + m * 256 + tid];
        }
    }
}

constexpr int QR = 512, QA = 1024, QP = 16, QG1 = 2 * QR + QA, QNW = 9 * QR + QG1;

struct BwdPlan {
    int G, nrb, jobsU, jobs2, jobs3;
    size_t lds_bytes, o_wt1, o_wt2, o_wt3, o_wt4, o_dgl, o_dz2, o_dg1, o_dp1, o_dga, o_p1, o_pg, o_dz, o_fm, o_wp, o_gemm, o_trace, total, gemm_bytes;
    ekaid_gemm_t gF[3], gWp;
};

bool bwd_plan(const ekaid_speaker_seq_t* d, void* workspace, BwdPlan& pl) {
    if (!(d->R == QR && d->A == QA && d->P == QP && d->G1 == QG1 && d->T >= 1 && d->T <= PMAX_T && d->B >= 1 && (d->We % 4) == 0)) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const int G = cus >= 256 ? 256 : (cus >= 128 ? 128 : 0);
    if (!G) return false;
    pl.G = G;
    pl.nrb = (d->B + 15) / 16;
    auto jobs = [&](int tiles) { return (tiles * pl.nrb + G - 1) / G; };
    pl.jobsU = jobs(QR / 16); pl.jobs2 = jobs(QA / 16); pl.jobs3 = jobs(QG1 / 16);
    if (pl.jobsU > MAXJ || pl.jobs2 > MAXJ || pl.jobs3 > MAXJ || (d->B + G - 1) / G > 4) return false;
    pl.lds_bytes = (size_t)(4 + 2 * 5120 + 96 + 32 + 16 + 12 + pl.jobsU * 1024 + pl.jobs2 * 1536 + pl.jobs3 * 1536) * sizeof(float);
    if (pl.lds_bytes > 160 * 1024) return false;
    if (!grid_fits(reinterpret_cast<const void*>(spk_seq_bwd_persistent_kernel<QR, QA, QP>), G, cus, pl.lds_bytes)) return false;
    const size_t Bp = 16 * (size_t)pl.nrb, f = sizeof(float);
    size_t o = al256((size_t)SYNC_WORDS * 4);
    pl.o_wt1 = o; o += al256((size_t)QA * 4 * QR * f);
    pl.o_wt2 = o; o += al256((size_t)QG1 * QA * f);
    pl.o_wt3 = o; o += al256((size_t)QR * 4 * QR * f);
    pl.o_wt4 = o; o += al256((size_t)QR * QNW * f);
    pl.o_dgl = o; o += al256(Bp * 4 * QR * f);
    pl.o_dz2 = o; o += al256(Bp * QA * f);
    pl.o_dg1 = o; o += al256(Bp * QG1 * f);
    pl.o_dp1 = o; o += al256(Bp * QR * f);
    pl.o_dga = o; o += al256(Bp * 4 * QR * f);
    pl.o_p1 = o; o += al256((size_t)pl.nrb * (QG1 / 16) * 16 * 20 * f);
    pl.o_pg = o; o += al256((size_t)pl.nrb * (QA / 16) * 16 * 4 * f);
    pl.o_dz = o; o += al256(Bp * 4 * f);
    pl.o_fm = o; o += al256(3 * (size_t)d->B * QG1 * f);
    pl.o_wp = o; o += al256((size_t)QG1 * QP * f);
    char* ws = (char*)workspace;
    float* Fm = ws ? (float*)(ws + pl.o_fm) : nullptr;
    const float* feats[3] = {d->bef, d->diff, d->aft};
    static const float dummy = 0.f;
    pl.gemm_bytes = 0;
    for (int m = 0; m < 3; ++m) {
        pl.gF[m] = dense_desc(feats[m] ? feats[m] : &dummy, QA, d->w_g1 ? d->w_g1 + 2 * QR : &dummy, QG1, 1, Fm ? Fm + (size_t)m * d->B * QG1 : nullptr, QG1, d->B, QG1, QA);
        pl.gemm_bytes = std::max(pl.gemm_bytes, ekaid_gemm_workspace_bytes(&pl.gF[m]));
    }
    pl.gWp = dense_desc(d->w_g1 ? d->w_g1 + QR : &dummy, QG1, d->w_p2 ? d->w_p2 : &dummy, QP, 0, ws ? (float*)(ws + pl.o_wp) : nullptr, QP, QG1, QP, QR);
    pl.gemm_bytes = std::max(pl.gemm_bytes, ekaid_gemm_workspace_bytes(&pl.gWp));
    pl.o_gemm = o; o += al256(pl.gemm_bytes);
    pl.o_trace = o;
    if (persist_trace_enabled()) o += al256((size_t)G * d->T * 16 * sizeof(unsigned long long));
    pl.total = o;
    return true;
}

}  // namespace
}  // namespace ekaid

using namespace ekaid;

extern "C" size_t ekaid_speaker_seq_bwd_persistent_workspace_bytes(const ekaid_speaker_seq_t* d) {
    BwdPlan pl;
    if (!d || !bwd_plan(d, nullptr, pl)) return 0;
    return pl.total;
}

extern "C" int ekaid_speaker_seq_bwd_persistent(const ekaid_speaker_seq_bwd_t* b, float* dFm, void* workspace, size_t workspace_bytes, void* stream) {
    if (!b || !dFm) { set_error("ekaid_speaker_seq_bwd_persistent: null descriptor / dFm"); return EKAID_ERR_ARG; }
    const ekaid_speaker_seq_t* d = &b->f;
    BwdPlan pl;
    if (!bwd_plan(d, workspace, pl)) {
        set_error("ekaid_speaker_seq_bwd_persistent: unsupported shape (R=%d A=%d P=%d G1=%d T=%d B=%d) or device", d->R, d->A, d->P, d->G1, d->T, d->B);
        return EKAID_ERR_UNSUPPORTED;
    }
    if (!workspace || workspace_bytes < pl.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        set_error("ekaid_speaker_seq_bwd_persistent: workspace too small or misaligned (%zu < %zu)", workspace_bytes, pl.total);
        return EKAID_ERR_ARG;
    }
    if (!d->seeds && (d->p_lm > 0.f || d->p_pos > 0.f)) { set_error("ekaid_speaker_seq_bwd_persistent: dropout without seeds"); return EKAID_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PBwd a;
    memset(&a, 0, sizeof(a));
    a.T = d->T; a.B = d->B; a.G = pl.G; a.nrb = pl.nrb;
    a.p_lm = d->p_lm; a.p_pos = d->p_pos;
    a.inv_lm = d->p_lm > 0.f ? 1.f / (1.f - d->p_lm) : 1.f;
    a.inv_pos = d->p_pos > 0.f ? 1.f / (1.f - d->p_pos) : 1.f;
    a.ep = dropout_epoch_ptr();
    a.WT1 = (const float*)(ws + pl.o_wt1); a.WT2 = (const float*)(ws + pl.o_wt2); a.WT3 = (const float*)(ws + pl.o_wt3); a.WT4 = (const float*)(ws + pl.o_wt4);
    a.w_fc = d->w_fc; a.w_wp = d->w_wp; a.bef = d->bef; a.diff = d->diff; a.aft = d->aft;
    a.Fm = (const float*)(ws + pl.o_fm); a.Wp = (const float*)(ws + pl.o_wp);
    a.CA = d->CA; a.CL = d->CL; a.SA = d->SA; a.SL = d->SL; a.G1D = d->G1D; a.MW = d->MW; a.VP = d->VP; a.PW = d->PW; a.AT = d->AT; a.GT = d->GT;
    a.dHLo = b->dHLo; a.dDPl = b->dDPl;
    a.DCAT = b->DCAT; a.DZFC = b->DZFC; a.DWP = b->DWP; a.DZ2 = b->DZ2; a.d_bef = b->d_bef; a.d_diff = b->d_diff; a.d_aft = b->d_aft; a.dFm = dFm;
    a.DGLfm = (float*)(ws + pl.o_dgl); a.DZ2fm = (float*)(ws + pl.o_dz2); a.DG1fm = (float*)(ws + pl.o_dg1); a.DP1fm = (float*)(ws + pl.o_dp1);
    a.DGAfm = (float*)(ws + pl.o_dga); a.PART1 = (float*)(ws + pl.o_p1); a.PARTG = (float*)(ws + pl.o_pg); a.DZ = (float*)(ws + pl.o_dz);
    a.sync = (unsigned*)workspace;
    a.status = seq_status_ptr();
    if (!a.status) { set_error("persistent sequence kernel: cannot allocate the status word"); return EKAID_ERR_LAUNCH; }
    a.trace = persist_trace_enabled() ? reinterpret_cast<unsigned long long*>(ws + pl.o_trace) : nullptr;
    a.jobsU = pl.jobsU; a.jobs2 = pl.jobs2; a.jobs3 = pl.jobs3;
    if (d->seeds) memcpy(a.seeds, d->seeds, sizeof(uint64_t) * 3 * d->T);
    const void* ptrs[] = {d->Wcat, d->whh_a, d->w_fc, d->w_wp, d->w_p2, d->w_g1, d->w_g2, d->wih_l, d->bef, d->diff, d->aft, d->CA, d->CL, d->SA, d->SL,
                          d->G1D, d->MW, d->VP, d->PW, d->AT, d->GT, b->dHLo, b->DCAT, b->DZFC, b->DWP, b->DZ2, b->d_bef, b->d_diff, b->d_aft, dFm};
    for (const void* p : ptrs)
        if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) { set_error("ekaid_speaker_seq_bwd_persistent: null or misaligned (16 B) pointer in the descriptor"); return EKAID_ERR_ARG; }
    // transposed weight copies: every contraction of the kernel reads its weight rows k-contiguously
    int rc;
    if ((rc = ekaid_transpose(d->wih_l + d->We, 4 * QR, QA, d->We + QA, (float*)(ws + pl.o_wt1), 4 * QR, stream))) return rc;
    if ((rc = ekaid_transpose(d->w_g2, QA, QG1, QG1, (float*)(ws + pl.o_wt2), QA, stream))) return rc;
    if ((rc = ekaid_transpose(d->whh_a, 4 * QR, QR, QR, (float*)(ws + pl.o_wt3), 4 * QR, stream))) return rc;
    if ((rc = ekaid_transpose(d->Wcat, QNW, QR, QR, (float*)(ws + pl.o_wt4), QNW, stream))) return rc;
    void* gws = ws + pl.o_gemm;
    for (int m = 0; m < 3; ++m)
        if ((rc = ekaid_gemm_f32(&pl.gF[m], gws, pl.gemm_bytes, stream))) return rc;
    if ((rc = ekaid_gemm_f32(&pl.gWp, gws, pl.gemm_bytes, stream))) return rc;
    if (hipMemsetAsync(workspace, 0, al256((size_t)SYNC_WORDS * 4), st) != hipSuccess) { (void)hipGetLastError(); set_error("ekaid_speaker_seq_bwd_persistent: memset failed"); return EKAID_ERR_LAUNCH; }
    auto kern = spk_seq_bwd_persistent_kernel<QR, QA, QP>;
    if (func_attr_once(reinterpret_cast<const void*>(kern))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("ekaid_speaker_seq_bwd_persistent: cannot raise the dynamic LDS limit");
            return EKAID_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(pl.G), dim3(256), pl.lds_bytes, st, a);
    count_launch(EKAID_COUNT_SPK_BWD_PERSISTENT);
    return check_launch("speaker_seq_bwd_persistent");
}
