// Sequence-level entry points: the time loops of the question GRU and of the teacher-forced answer decoder issued from C.
// Each step is the same chain of C-ABI calls the Python nodes made one by one (slab products + the kernels that consume the
// slabs, csrc/steps.hip); nothing here allocates or synchronises, so a whole sequence is still graph-capturable.  The
// workspace holds the K-slabs of ONE step (every product of a step has its own region; the next step reuses them, ordered
// by the stream) and the small carries of the backward loops.
// ref: language_model.py:106-115 (nn.GRU), dynamic_speaker_change_pos.py:94-131,196-222 (DynamicCore inside _forward).
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <cstdlib>
#include <cstring>

namespace ekaid {
namespace {

struct Product {            // one slab product of a step: descriptor, slab count, region inside the workspace
    ekaid_gemm_t g;
    int n;
    float* buf;
    size_t bytes() const { return (size_t)n * g.M * g.N * sizeof(float); }
    ekaid_slabs_t cols(int c0) const { return ekaid_slabs_t{buf + c0, n, (long long)g.M * g.N, g.N}; }
};

// op(A) . op(B) with A = M x K (row-major, leading dim lda) and B stored [N][K] (tb = 1) or [K][N] (tb = 0)
Product product(const float* A, int lda, const float* B, int ldb, int tb, int M, int N, int K) {
    static const float dummy = 0.f;         // planning only looks at the shape; the step loops fill in the operands
    Product p;
    memset(&p.g, 0, sizeof(p.g));
    p.g.A = A ? A : &dummy; p.g.B = B ? B : &dummy; p.g.C = nullptr;
    p.g.M = M; p.g.N = N; p.g.K = K;
    p.g.lda = lda; p.g.ldb = ldb; p.g.ldc = N;
    p.g.transA = 0; p.g.transB = tb;
    p.g.batch1 = p.g.batch2 = 1;
    p.g.alpha = 1.f;
    p.g.rdiv = 1;
    p.n = ekaid_gemm_slab_count(&p.g);
    p.buf = nullptr;
    return p;
}

struct Carver {             // hands out 256-byte aligned regions of the caller's workspace (or only adds up their sizes)
    char* base; size_t used;
    float* take(size_t bytes) {
        float* p = base ? reinterpret_cast<float*>(base + used) : nullptr;
        used += (bytes + 255) & ~(size_t)255;
        return p;
    }
};

const ekaid_slabs_t NO_SLABS = {nullptr, 0, 0, 0};

int run(const Product& p, void* stream) { return ekaid_gemm_slabs(&p.g, p.buf, p.bytes(), stream); }
int run2(const Product& p, const Product& q, void* stream) {
    return ekaid_gemm_slabs2(&p.g, p.buf, p.bytes(), &q.g, q.buf, q.bytes(), stream);
}

// ---------------------------------------------------------------- GRU
struct GruPlan { Product gh, carry; float* dh[2]; size_t bytes; };

bool gru_plan(int B, int T, int H, const float* w_hh, void* ws, bool backward, GruPlan& pl) {
    (void)T;
    Carver c{(char*)ws, 0};
    pl.gh = product(nullptr, H, w_hh, H, 1, B, 3 * H, H);          // h_{t-1} . W_hh^T
    pl.carry = product(nullptr, 3 * H, w_hh, H, 0, B, H, 3 * H);    // dgh_t . W_hh
    if (pl.gh.n < 1 || pl.carry.n < 1) return false;
    if (!backward) pl.gh.buf = c.take(pl.gh.bytes());
    else {
        pl.carry.buf = c.take(pl.carry.bytes());
        pl.dh[0] = c.take((size_t)B * H * sizeof(float));
        pl.dh[1] = c.take((size_t)B * H * sizeof(float));
    }
    pl.bytes = c.used;
    return true;
}

// ---------------------------------------------------------------- answer decoder
struct SpkPlan {
    Product p0, p1, p2, p3, p4;             // forward products of a step
    Product q1, q2, q3, ca, cl;             // backward products of a step
    float *dc_l, *dc_a, *dc_tmp, *d_gated;
    size_t bytes;
};

bool spk_plan(const ekaid_speaker_seq_t* d, void* ws, bool backward, SpkPlan& pl) {
    const int B = d->B, R = d->R, A = d->A, G1 = d->G1, NW = 9 * R + G1, We = d->We;
    Carver c{(char*)ws, 0};
    if (!backward) {
        pl.p0 = product(nullptr, R, d->Wcat, R, 1, B, NW, R);                 // h_lang_{t-1} . Wcat^T
        pl.p1 = product(nullptr, R, d->whh_a, R, 1, B, 4 * R, R);             // h_att_{t-1} . W_hh_att^T
        pl.p2 = product(nullptr, R + A, d->w_g1 + R, G1, 1, B, G1, R + A);    // [pos_feat | att] . W_g1[:, R:]^T
        pl.p3 = product(nullptr, G1, d->w_g2, G1, 1, B, A, G1);               // g1d . W_g2^T
        pl.p4 = product(nullptr, A, d->wih_l + We, We + A, 1, B, 4 * R, A);   // gated . W_ih_lang[:, We:]^T
        Product* ps[5] = {&pl.p0, &pl.p1, &pl.p2, &pl.p3, &pl.p4};
        for (Product* p : ps) {
            if (p->n < 1) return false;
            p->buf = c.take(p->bytes());
        }
    } else {
        pl.q1 = product(nullptr, NW, d->wih_l + We, We + A, 0, B, A, 4 * R);  // d lang gates . W_ih_lang[:, We:]
        pl.q2 = product(nullptr, A, d->w_g2, G1, 0, B, G1, A);                // d z2 . W_g2
        pl.q3 = product(nullptr, NW, d->w_g1 + R, G1, 0, B, R + A, G1);       // d gate1x . W_g1[:, R:]
        pl.ca = product(nullptr, NW, d->whh_a, R, 0, B, R, 4 * R);            // d att gates . W_hh_att
        pl.cl = product(nullptr, NW, d->Wcat, R, 0, B, R, NW);                // DCAT_t . Wcat
        Product* ps[5] = {&pl.q1, &pl.q2, &pl.q3, &pl.ca, &pl.cl};
        for (Product* p : ps) {
            if (p->n < 1) return false;
            p->buf = c.take(p->bytes());
        }
        pl.dc_l = c.take((size_t)B * R * sizeof(float));
        pl.dc_a = c.take((size_t)B * R * sizeof(float));
        pl.dc_tmp = c.take((size_t)B * R * sizeof(float));
        pl.d_gated = c.take((size_t)B * A * sizeof(float));
    }
    pl.bytes = c.used;
    return true;
}

bool spk_args_ok(const ekaid_speaker_seq_t* d, const char* who) {
    if (!d) { set_error("%s: null descriptor", who); return false; }
    if (d->T < 1 || d->B < 1 || d->R < 1 || d->A < 1 || d->P < 1 || d->G1 < 1 || d->We < 0) {
        set_error("%s: bad dims T=%d B=%d R=%d A=%d P=%d G1=%d We=%d", who, d->T, d->B, d->R, d->A, d->P, d->G1, d->We);
        return false;
    }
    if (!d->seeds && (d->p_lm > 0.f || d->p_pos > 0.f)) { set_error("%s: dropout without seeds", who); return false; }
    return true;
}

}  // namespace
}  // namespace ekaid

using namespace ekaid;

#define EKAID_TRY(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

extern "C" size_t ekaid_gru_seq_workspace_bytes(int B, int T, int H, int backward) {
    if (B < 1 || T < 1 || H < 1) return 0;
    GruPlan pl;
    if (!gru_plan(B, T, H, nullptr, nullptr, backward != 0, pl)) return 0;
    const size_t pers = ekaid_gru_seq_persistent_workspace_bytes(B, T, H, backward);      // either form fits
    return pl.bytes > pers ? pl.bytes : pers;
}

extern "C" int ekaid_gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, int B, int T, int H, float* hall, float* save,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!gi || !w_hh || !b_hh || !hall || !save || B < 1 || T < 1 || H < 1) { set_error("ekaid_gru_seq_fwd: bad arguments"); return EKAID_ERR_ARG; }
    // one persistent launch where the shape allows it AND it measured faster (csrc/persist_gru.hip; tools/gru_bench.py, same box:
    // 12.6 -> 10.7 us per step at 32 rows, 13.8 -> 10.2 at 64, 16.3 -> 14.6 at 128, level at 256)
    if (seq_mode() == 1 && T >= 4 && B <= 128) {
        const size_t pers = ekaid_gru_seq_persistent_workspace_bytes(B, T, H, 0);
        if (pers && workspace && workspace_bytes >= pers) return ekaid_gru_seq_fwd_persistent(gi, w_hh, b_hh, B, T, H, hall, save, workspace, workspace_bytes, stream);
    }
    GruPlan pl;
    if (!gru_plan(B, T, H, w_hh, workspace, false, pl)) return EKAID_ERR_ARG;
    if (!workspace || workspace_bytes < pl.bytes) { set_error("ekaid_gru_seq_fwd: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes); return EKAID_ERR_ARG; }
    const ekaid_slabs_t gh = pl.gh.cols(0);
    for (int t = 0; t < T; ++t) {
        const float* h = hall + (size_t)t * B * H;
        pl.gh.g.A = h;
        EKAID_TRY(run(pl.gh, stream));
        EKAID_TRY(ekaid_gru_cell_slabs_fwd(gi + (size_t)t * 3 * H, T * 3 * H, &gh, b_hh, h, H, B, H, hall + (size_t)(t + 1) * B * H, H,
                                           save + (size_t)t * B * 4 * H, stream));
    }
    return EKAID_OK;
}

extern "C" int ekaid_gru_seq_bwd(const float* dout, const float* w_hh, const float* hall, const float* save, int B, int T, int H,
                                 float* dgi, float* dgh, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dout || !w_hh || !hall || !save || !dgi || !dgh || B < 1 || T < 1 || H < 1) { set_error("ekaid_gru_seq_bwd: bad arguments"); return EKAID_ERR_ARG; }
    // the persistent backward is NOT routed by default: one seam + a K = 3H carry job + the element-wise cell backward per step measured
    // 14.5-14.8 us against 11.9-12.8 us for the two small launches of the step-wise chain at 32-64 rows, 37 against 22-24 at 256
    // (tools/gru_bench.py); EKAID_GRU_BWD=persistent selects it (tests call the entry directly)
    static const bool gru_bwd_persistent = [] { const char* e = getenv("EKAID_GRU_BWD"); return e && e[0] == 'p'; }();
    if (seq_mode() == 1 && T >= 4 && gru_bwd_persistent) {
        const size_t pers = ekaid_gru_seq_persistent_workspace_bytes(B, T, H, 1);
        if (pers && workspace && workspace_bytes >= pers) return ekaid_gru_seq_bwd_persistent(dout, w_hh, hall, save, B, T, H, dgi, dgh, workspace, workspace_bytes, stream);
    }
    GruPlan pl;
    if (!gru_plan(B, T, H, w_hh, workspace, true, pl)) return EKAID_ERR_ARG;
    if (!workspace || workspace_bytes < pl.bytes) { set_error("ekaid_gru_seq_bwd: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes); return EKAID_ERR_ARG; }
    const ekaid_slabs_t carry = pl.carry.cols(0);
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        EKAID_TRY(ekaid_gru_cell_slabs_bwd(dout + (size_t)t * H, T * H, last ? nullptr : pl.dh[(t + 1) & 1], last ? &NO_SLABS : &carry,
                                           save + (size_t)t * B * 4 * H, hall + (size_t)t * B * H, H, B, H, dgi + (size_t)t * 3 * H,
                                           T * 3 * H, dgh + (size_t)t * B * 3 * H, pl.dh[t & 1], stream));
        if (t > 0) {        // dh_{t-1} = z * d (dh of this step) + dgh_t . W_hh (slabs, summed by the next cell kernel)
            pl.carry.g.A = dgh + (size_t)t * B * 3 * H;
            EKAID_TRY(run(pl.carry, stream));
        }
    }
    return EKAID_OK;
}

extern "C" size_t ekaid_speaker_seq_workspace_bytes(const ekaid_speaker_seq_t* d, int backward) {
    if (!spk_args_ok(d, "ekaid_speaker_seq_workspace_bytes")) return 0;
    SpkPlan pl;
    if (!spk_plan(d, nullptr, backward != 0, pl)) return 0;
    // large enough for either sequence mode (the persistent kernel only needs its seam counters)
    const size_t pers = backward ? 0 : ekaid_speaker_seq_persistent_workspace_bytes(d);
    return pl.bytes > pers ? pl.bytes : pers;
}

extern "C" int ekaid_speaker_seq_fwd(const ekaid_speaker_seq_t* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (!spk_args_ok(d, "ekaid_speaker_seq_fwd")) return EKAID_ERR_ARG;
    if (seq_mode() == 1) {                  // one persistent launch for the whole sequence where the shape allows it (csrc/persist.hip); short
                                            // sequences do not amortise its prologue (5 small products, the layout conversion of the initial states)
        const size_t pers = ekaid_speaker_seq_persistent_workspace_bytes(d);
        if (pers && d->T >= 8 && workspace && workspace_bytes >= pers) return ekaid_speaker_seq_fwd_persistent(d, workspace, workspace_bytes, stream);
    }
    SpkPlan pl;
    if (!spk_plan(d, workspace, false, pl)) return EKAID_ERR_ARG;
    if (!workspace || workspace_bytes < pl.bytes) { set_error("ekaid_speaker_seq_fwd: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes); return EKAID_ERR_ARG; }
    count_launch(EKAID_COUNT_SPK_FWD_STEPWISE);
    const int T = d->T, B = d->B, R = d->R, A = d->A, P = d->P, G1 = d->G1;
    const int oP1 = 4 * R, oG1 = 5 * R, oGL = 5 * R + G1;
    const size_t BR = (size_t)B * R;
    ekaid_spk_step_a_fwd_t sa;
    memset(&sa, 0, sizeof(sa));
    sa.B = B; sa.R = R; sa.A = A; sa.P = P;
    sa.b_hh = d->bhh_a; sa.w_fc = d->w_fc; sa.b_fc = d->b_fc;
    sa.bef = d->bef; sa.diff = d->diff; sa.aft = d->aft;
    sa.b_p1 = d->b_p1; sa.w_wp = d->w_wp; sa.b_wp = d->b_wp; sa.w_p2 = d->w_p2; sa.b_p2 = d->b_p2;
    sa.p_v = d->p_lm; sa.p_d = d->p_pos; sa.ld2 = R + A; sa.ldpp = R + A; sa.n_gates_src = 2;
    sa.gates_src[0] = pl.p0.cols(0);
    sa.gates_src[1] = pl.p1.cols(0);
    sa.pos_src = pl.p0.cols(oP1);
    const ekaid_slabs_t g1_src[2] = {pl.p0.cols(oG1), pl.p2.cols(0)};
    const ekaid_slabs_t gl_src[2] = {pl.p0.cols(oGL), pl.p4.cols(0)};
    const ekaid_slabs_t p3 = pl.p3.cols(0);
    for (int t = 0; t < T; ++t) {
        const size_t o_prev = t * BR, o_t = (t + 1) * BR;
        const uint64_t sv = d->seeds ? d->seeds[3 * t] : 0, sd = d->seeds ? d->seeds[3 * t + 1] : 0, sg = d->seeds ? d->seeds[3 * t + 2] : 0;
        float* pa_t = d->PA + (size_t)t * B * (R + A);
        // everything h_lang_{t-1} feeds (NW columns) and h_att_{t-1} . W_hh_att^T: two products, one launch
        pl.p0.g.A = d->HL + o_prev;
        pl.p1.g.A = d->HA + o_prev;
        EKAID_TRY(run2(pl.p0, pl.p1, stream));
        // att-LSTM cell -> module weights + attended feature -> POS chain: one row-local kernel
        sa.gxa = d->GXA + (size_t)t * B * 4 * R;
        sa.c_prev = d->CA + o_prev;
        sa.h_out = d->HA + o_t; sa.c_out = d->CA + o_t;
        sa.gate_save = d->SA + (size_t)t * B * 4 * R;
        sa.mw = d->MW + (size_t)t * B * 3;
        sa.att = d->AT + (size_t)t * B * A;
        sa.att2 = pa_t + R;
        sa.seed_v = sv; sa.seed_d = sd;
        sa.vp = d->VP + t * BR; sa.dp = d->DP + (size_t)t * B * P; sa.pw = d->PW + (size_t)t * B * P;
        sa.pp = pa_t;
        EKAID_TRY(ekaid_spk_step_a_fwd(&sa, stream));
        // gate MLP, layer 1: dropout(relu(h_lang block + [pos_feat | att] W_g1[:, R:]^T + b))
        pl.p2.g.A = pa_t;
        EKAID_TRY(run(pl.p2, stream));
        EKAID_TRY(ekaid_slab_epilogue(g1_src, 2, d->b_g1, B, G1, ACT_RELU, d->p_lm, sg, d->G1D + (size_t)t * B * G1, G1, stream));
        // layer 2 + gating: gate = sigmoid(. W_g2^T + b); gated = gate * att
        pl.p3.g.A = d->G1D + (size_t)t * B * G1;
        EKAID_TRY(run(pl.p3, stream));
        EKAID_TRY(ekaid_gate_mul_slabs_fwd(&p3, d->b_g2, d->AT + (size_t)t * B * A, B, A, d->GT + (size_t)t * B * A,
                                           d->GA + (size_t)t * B * A, stream));
        // language LSTM: gates = GXL[t] + h_lang block + gated W_ih[:, We:]^T + b_hh
        pl.p4.g.A = d->GA + (size_t)t * B * A;
        EKAID_TRY(run(pl.p4, stream));
        EKAID_TRY(ekaid_lstm_cell_slabs_fwd(d->GXL + (size_t)t * B * 4 * R, gl_src, 2, d->bhh_l, d->CL + o_prev, B, R, d->HL + o_t,
                                            d->CL + o_t, d->SL + (size_t)t * B * 4 * R, stream));
    }
    return EKAID_OK;
}

extern "C" int ekaid_speaker_seq_bwd(const ekaid_speaker_seq_bwd_t* b, void* workspace, size_t workspace_bytes, void* stream) {
    if (!b) { set_error("ekaid_speaker_seq_bwd: null descriptor"); return EKAID_ERR_ARG; }
    const ekaid_speaker_seq_t* d = &b->f;
    if (!spk_args_ok(d, "ekaid_speaker_seq_bwd")) return EKAID_ERR_ARG;
    SpkPlan pl;
    if (!spk_plan(d, workspace, true, pl)) return EKAID_ERR_ARG;
    if (!workspace || workspace_bytes < pl.bytes) { set_error("ekaid_speaker_seq_bwd: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes); return EKAID_ERR_ARG; }
    count_launch(EKAID_COUNT_SPK_BWD_STEPWISE);
    const int T = d->T, B = d->B, R = d->R, A = d->A, P = d->P, G1 = d->G1, NW = 9 * R + G1;
    const int oP1 = 4 * R, oG1 = 5 * R, oGL = 5 * R + G1;
    const size_t BR = (size_t)B * R;
    hipStream_t st = (hipStream_t)stream;
    // the cell-state carries start at zero (dc_l and dc_a are adjacent regions of the workspace)
    if (hipMemsetAsync(pl.dc_l, 0, (size_t)((char*)pl.dc_tmp - (char*)pl.dc_l), st) != hipSuccess) {
        (void)hipGetLastError();
        set_error("ekaid_speaker_seq_bwd: memset failed");
        return EKAID_ERR_LAUNCH;
    }
    float *dc_l = pl.dc_l, *dc_a = pl.dc_a, *dc_tmp = pl.dc_tmp;
    ekaid_spk_step_a_bwd_t sb;
    memset(&sb, 0, sizeof(sb));
    sb.B = B; sb.R = R; sb.A = A; sb.P = P;
    sb.d_gated = pl.d_gated; sb.bef = d->bef; sb.diff = d->diff; sb.aft = d->aft; sb.w_fc = d->w_fc;
    sb.d_bef = b->d_bef; sb.d_diff = b->d_diff; sb.d_aft = b->d_aft;
    sb.w_p2 = d->w_p2; sb.w_wp = d->w_wp;
    sb.p_v = d->p_lm; sb.p_d = d->p_pos; sb.lddg = NW; sb.lddpp = R; sb.lddx = NW;
    sb.dpa_src = pl.q3.cols(0);
    const ekaid_slabs_t s1 = pl.q1.cols(0), s2 = pl.q2.cols(0), carry_l = pl.cl.cols(0), carry_a = pl.ca.cols(0);
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        const size_t o_prev = t * BR, o_t = (t + 1) * BR;
        float* dcat = b->DCAT + (size_t)t * B * NW;
        const uint64_t sv = d->seeds ? d->seeds[3 * t] : 0, sd = d->seeds ? d->seeds[3 * t + 1] : 0, sg = d->seeds ? d->seeds[3 * t + 2] : 0;
        // language LSTM (upstream on h_lang_t = carry slabs + output path)
        EKAID_TRY(ekaid_lstm_cell_slabs_bwd(last ? &NO_SLABS : &carry_l, b->dHLo + t * BR, dc_l, d->SL + (size_t)t * B * 4 * R,
                                            d->CL + o_prev, d->CL + o_t, B, R, dcat + oGL, NW, dc_tmp, stream));
        { float* x = dc_l; dc_l = dc_tmp; dc_tmp = x; }
        // d gated = d lang gates . W_ih_lang[:, We:]
        pl.q1.g.A = dcat + oGL;
        EKAID_TRY(run(pl.q1, stream));
        // gated = gate * att ; gate = sigmoid(W_g2 g1d + b):  d z2 = d gated * att * gate'
        EKAID_TRY(ekaid_gate_slabs_bwd(&s1, d->AT + (size_t)t * B * A, d->GT + (size_t)t * B * A, B, A, b->DZ2 + (size_t)t * B * A,
                                       pl.d_gated, stream));
        pl.q2.g.A = b->DZ2 + (size_t)t * B * A;
        EKAID_TRY(run(pl.q2, stream));
        EKAID_TRY(ekaid_drop_act_slabs_bwd(&s2, d->G1D + (size_t)t * B * G1, G1, B, G1, ACT_RELU, d->p_lm, sg, dcat + oG1, NW, stream));
        // [d pos_feat | d att (gate-MLP path)] = d gate1x . W_g1[:, R:]
        pl.q3.g.A = dcat + oG1;
        EKAID_TRY(run(pl.q3, stream));
        // module mix, att-LSTM cell, POS chain: one row-local kernel
        sb.gate = d->GT + (size_t)t * B * A;
        sb.mw = d->MW + (size_t)t * B * 3;
        sb.dz_fc = b->DZFC + (size_t)t * B * 3;
        sb.dh_carry = last ? NO_SLABS : carry_a;
        sb.dc_in = dc_a; sb.dc_out = dc_tmp;
        sb.gate_save = d->SA + (size_t)t * B * 4 * R;
        sb.c_prev = d->CA + o_prev; sb.c_cur = d->CA + o_t;
        sb.dgates = dcat;
        sb.pw = d->PW + (size_t)t * B * P;
        sb.dhead = b->dDPl ? b->dDPl + (size_t)t * B * P : nullptr;
        sb.vp = d->VP + t * BR;
        sb.seed_v = sv; sb.seed_d = sd;
        sb.dpp_out = b->DPP + t * BR;
        sb.dwp_out = b->DWP + (size_t)t * B * P;
        sb.dx_out = dcat + oP1;
        EKAID_TRY(ekaid_spk_step_a_bwd(&sb, stream));
        { float* x = dc_a; dc_a = dc_tmp; dc_tmp = x; }
        if (t > 0) {    // what flowed out of h_att_{t-1} (att gates . W_hh) and out of h_lang_{t-1} (everything . Wcat): one launch
            pl.ca.g.A = dcat;
            pl.cl.g.A = dcat;
            EKAID_TRY(run2(pl.ca, pl.cl, stream));
        }
    }
    return EKAID_OK;
}

// ---------------------------------------------------------------- greedy decoding (sample_max = 1), eval mode
// Steps t0 .. t1-1 of DynamicSpeaker._sample's loop without a host round trip per step: token embedding -> word-side gates of
// the language LSTM -> the recurrent step (ekaid_speaker_seq_fwd on the rows of step t) -> logits -> log-softmax -> arg-max /
// stop-flag bookkeeping (ekaid_greedy_step), the chosen tokens feeding the next step on the device.  The host polls
// n_unfinished every few steps (the reference tests "all finished" every step, dynamic_speaker_change_pos.py:354).
namespace ekaid {
namespace {
constexpr int PERSISTENT_GREEDY_MAX_STEPS = 96;
struct GreedyPlan { ekaid_gemm_t gxl, lo; float *ws_gxl, *ws_lo; void* ws_seq; size_t seq_bytes, bytes; };

ekaid_gemm_t dense_nt(const float* A, int lda, const float* W, int ldw, float* C, int M, int N, int K, const float* bias) {
    ekaid_gemm_t g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldw; g.ldc = N;
    g.transA = 0; g.transB = 1; g.batch1 = g.batch2 = 1; g.alpha = 1.f; g.rdiv = 1; g.bias = bias;
    return g;
}

bool greedy_plan(const ekaid_speaker_greedy_t* q, void* ws, GreedyPlan& pl) {
    const ekaid_speaker_seq_t* d = &q->f;
    const int B = d->B, R = d->R, A = d->A, We = d->We;
    pl.gxl = dense_nt(q->xt, We, d->wih_l, We + A, const_cast<float*>(d->GXL), B, 4 * R, We, q->bih_l);
    pl.lo = dense_nt(nullptr, R, q->w_lo, R, q->logits, B, q->V, R, q->b_lo);
    Carver c{(char*)ws, 0};
    pl.ws_gxl = c.take(ekaid_gemm_workspace_bytes(&pl.gxl));
    pl.ws_lo = c.take(ekaid_gemm_workspace_bytes(&pl.lo));
    ekaid_speaker_seq_t one = *d;
    one.T = 1;
    pl.seq_bytes = ekaid_speaker_seq_workspace_bytes(&one, 0);
    if (pl.seq_bytes == 0) return false;
    pl.ws_seq = c.take(pl.seq_bytes);
    pl.bytes = c.used;
    return true;
}
}  // namespace
}  // namespace ekaid

extern "C" size_t ekaid_speaker_greedy_workspace_bytes(const ekaid_speaker_greedy_t* q) {
    if (!q || !spk_args_ok(&q->f, "ekaid_speaker_greedy_workspace_bytes") || q->V < 1) return 0;
    GreedyPlan pl;
    if (!greedy_plan(q, nullptr, pl)) return 0;
    const size_t pers = ekaid_speaker_greedy_persistent_workspace_bytes(q, q->f.T < PERSISTENT_GREEDY_MAX_STEPS ? q->f.T : PERSISTENT_GREEDY_MAX_STEPS);
    return pl.bytes > pers ? pl.bytes : pers;
}

extern "C" int ekaid_speaker_greedy_steps(const ekaid_speaker_greedy_t* q, int t0, int t1, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (!q || !spk_args_ok(&q->f, "ekaid_speaker_greedy_steps")) return EKAID_ERR_ARG;
    const ekaid_speaker_seq_t* d = &q->f;
    if (d->p_lm != 0.f || d->p_pos != 0.f) { set_error("ekaid_speaker_greedy_steps: eval mode only (dropout probabilities must be 0)"); return EKAID_ERR_UNSUPPORTED; }
    if (q->V < 1 || t0 < 0 || t1 > d->T || t0 >= t1 || q->seq_len < t1 || !q->emb || !q->bih_l || !q->w_lo || !q->b_lo || !q->xt || !q->logits ||
        !q->logp || !q->it || !q->unfinished || !q->seq || !q->seq_logp || !q->n_unfinished || !d->GXL || !d->GXA) {
        set_error("ekaid_speaker_greedy_steps: bad arguments (steps [%d, %d) of %d)", t0, t1, d->T);
        return EKAID_ERR_ARG;
    }
    if (seq_mode() == 1 && t1 - t0 >= 4) {      // the whole range as ONE persistent launch where the shape allows it (csrc/persist.hip)
        const size_t pers = ekaid_speaker_greedy_persistent_workspace_bytes(q, t1 - t0);
        if (pers && workspace && workspace_bytes >= pers) return ekaid_speaker_greedy_steps_persistent(q, t0, t1, workspace, workspace_bytes, stream);
    }
    GreedyPlan pl;
    if (!greedy_plan(q, workspace, pl)) return EKAID_ERR_ARG;
    if (!workspace || workspace_bytes < pl.bytes) { set_error("ekaid_speaker_greedy_steps: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes); return EKAID_ERR_ARG; }
    count_launch(EKAID_COUNT_GREEDY_STEPWISE);
    const int B = d->B, R = d->R, We = d->We;
    const size_t BR = (size_t)B * R;
    for (int t = t0; t < t1; ++t) {
        // word side of the language LSTM: relu(emb[it]) . W_ih[:, :We]^T + b_ih   (:229, eval: no dropout)
        EKAID_TRY(ekaid_embedding_fwd(q->it, B, q->emb, We, q->xt, We, 0, 1, stream));
        EKAID_TRY(ekaid_gemm_f32(&pl.gxl, pl.ws_gxl, ekaid_gemm_workspace_bytes(&pl.gxl), stream));
        // the recurrent step on the state rows t -> t + 1 (the per-step activation buffers are reused, the module weights kept)
        ekaid_speaker_seq_t one = *d;
        one.T = 1;
        one.seeds = nullptr;
        one.HA = d->HA + t * BR; one.CA = d->CA + t * BR; one.HL = d->HL + t * BR; one.CL = d->CL + t * BR;
        one.MW = d->MW + (size_t)t * B * 3;
        EKAID_TRY(ekaid_speaker_seq_fwd(&one, pl.ws_seq, pl.seq_bytes, stream));
        // logits -> log-softmax -> arg-max, stop flags, next tokens (:238-239, :316-352)
        pl.lo.A = d->HL + (t + 1) * BR;
        EKAID_TRY(ekaid_gemm_f32(&pl.lo, pl.ws_lo, ekaid_gemm_workspace_bytes(&pl.lo), stream));
        EKAID_TRY(ekaid_softmax_fwd(q->logits, q->logp, B, q->V, 1, stream));
        EKAID_TRY(ekaid_greedy_step(q->logp, B, q->V, t, q->unfinished, q->it, q->seq, q->seq_logp, q->seq_len, q->n_unfinished + t, stream));
    }
    return EKAID_OK;
}
