// Persistent question GRU: the 20 recurrent steps of nn.GRU (1 layer, gate order r, z, n; language_model.py:106-115) in ONE launch
// per direction.  Same machinery as the decoder kernels (csrc/persist.h: 16 x 16 f32-MFMA jobs with the full contraction, fragment-
// major hand-offs behind device-scope counter seams), but a GRU step has ONE all-to-all dependency (h_t -> every gate of step t+1), so
// a step costs one seam + one job instead of two launches:
//   forward  job (16 hidden units x 16 rows): the three gate tiles r, z, n share the h_{t-1} fragments (weights W_hh resident in
//            registers for all steps' jobs of the phase); cell in the epilogue; h_t handed on fragment-major (double-buffered: it is
//            read and rewritten in the same phase); the previous state of the owned elements stays in LDS;
//   backward cell backward element-wise by the owner of (16 units x 16 rows) -> d gh_t (3H wide) handed on -> [seam] -> the carry
//            d h_{t-1} += d gh_t . W_hh (K = 3H, transposed weight copy made once per call) by the same owners, kept in LDS together
//            with the direct term z * d.
// Contracts of ekaid_gru_seq_fwd / ekaid_gru_seq_bwd (include/ekaid_hip.h) unchanged; H = 1024 only (anything else: step-wise chain).
#include "../../include/ekaid_hip.h"
#include "persist.h"

namespace ekaid {
namespace {

struct PGru {
    int T, B, G, nrb, jobs;
    const float *gi, *w, *b_hh;          // forward: gi [B][T][3H], w = W_hh [3H][H];  backward: w = W_hh^T [H][3H]
    float *hall, *save;                  // [T+1][B][H], [T][B][4H]
    const float* dout;                   // backward: [B][T][H]
    float *dgi, *dgh;                    // backward: [B][T][3H], [T][B][3H]
    float *Hfm /* fwd: [2][Bp][H] */, *Dfm /* bwd: [2][Bp][3H] */;
    unsigned* sync;
    unsigned* status;                 // per-device sticky word: non-zero = a seam timed out (api.hip: seq_status_ptr)
};

struct Pre3g { float r, z, n; };
struct PreNoneG { float x; };

template <int H>
__global__ __launch_bounds__(256, 1) void gru_seq_fwd_persistent_kernel(const PGru a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = H / 64, KB = H / 16, TILES = H / 16;
    constexpr int L_red = 4, L_fixed = L_red + 2 * 5120;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int B = a.B, G = a.G, nrb = a.nrb, T = a.T;
    const int l15 = lane & 15, kq4 = (lane >> 4) * 4, row = tid >> 4, col = tid & 15;
    const unsigned shard = xcc_id();
    unsigned* const status = a.status;
    Ctx c = {tid, lane, wave, wg, G, nrb, B, row, col, lds + L_red, 0, status, reinterpret_cast<int*>(lds), nullptr};
    float* const hstate = lds + L_fixed;                        // per job: h_{t-1} of the owned (row, unit) elements
    const long long Bp = 16ll * nrb, BH = (long long)B * H;
    const int tU = wg % TILES, rb0 = wg / TILES, drb = G / TILES;
    const int nj = wg < TILES * nrb ? (TILES * nrb - wg + G - 1) / G : 0;
    const int unit = 16 * tU + col;
    const float br = a.b_hh[unit], bz = a.b_hh[H + unit], bn = a.b_hh[2 * H + unit];
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.w, 3ll * H * H * 4);
    const unsigned kw = (unsigned)(wave * (H / 4) + kq4) * 4u;
    for (int n = 0; n < nj; ++n) hstate[n * 256 + tid] = a.hall[(long long)min((rb0 + n * drb) * 16 + row, B - 1) * H + unit];
    // prologue: h_0 into fragment-major layout (buffer 0), then the first seam
    {
        const int kq = tid >> 6, rr = (tid >> 2) & 15, e = tid & 3;
        for (int ch = wg; ch < nrb * KB; ch += G) {
            const int rb = ch / KB, kb = ch - rb * KB, gr = rb * 16 + rr;
            store_sc1(a.Hfm + ((long long)ch << 8) + tid, gr < B ? a.hall[(long long)gr * H + kb * 16 + kq * 4 + e] : 0.f);
        }
        seam_arrive(a.sync, shard);
        __syncthreads();
    }
    for (int t = 0; t < T; ++t) {
        const __amdgpu_buffer_rsrc_t rsR = make_rsrc(a.Hfm + (long long)(t & 1) * Bp * H, Bp * H * 4), rsWr = make_rsrc(a.Hfm + (long long)((t + 1) & 1) * Bp * H, Bp * H * 4);
        if (!run_mt<NB, 3, Pre3g>(
                c, TILES, [&](int m, int i) { return bload4<0>(rsW, (unsigned)((m * H + 16 * tU + l15) * H) * 4u + kw + 64u * i); }, rsR, KB, a.sync,
                (unsigned)G * (unsigned)(t + 1), 0x300u, -1,
                [&](int rb, int) {
                    const float* g = a.gi + ((long long)min(rb * 16 + row, B - 1) * T + t) * 3 * H + unit;
                    return Pre3g{g[0], g[H], g[2 * H]};
                },
                [&](const float (&v)[3], const Pre3g& gi, int rb, int n) {
                    const int grow = rb * 16 + row;
                    const float ghn = v[2] + bn;
                    const float r = sigm(gi.r + v[0] + br), z = sigm(gi.z + v[1] + bz);
                    const float nn = tanhf(gi.n + r * ghn);
                    const float hp = hstate[n * 256 + tid];
                    const float hn = (1.f - z) * nn + z * hp;
                    hstate[n * 256 + tid] = hn;
                    if (grow < B) {
                        a.hall[(long long)(t + 1) * BH + (long long)grow * H + unit] = hn;
                        float* sp = a.save + ((long long)t * B + grow) * 4 * H;
                        sp[unit] = r; sp[H + unit] = z; sp[2 * H + unit] = nn; sp[3 * H + unit] = ghn;
                    }
                    const f32x4 q = gather4(hn);
                    if ((col & 3) == 0) bstore4_sc1(rsWr, fm_chunk(rb, tU, KB) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
                }))
            return;
        seam_arrive(a.sync, shard);
    }
}

template <int H>
__global__ __launch_bounds__(256, 1) void gru_seq_bwd_persistent_kernel(const PGru a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = 3 * H / 64, KB = 3 * H / 16, TILES = H / 16, TG = H / 16;      // K = 3H: 48 blocks per wave
    constexpr int L_red = 4, L_fixed = L_red + 2 * 5120;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int B = a.B, G = a.G, nrb = a.nrb, T = a.T;
    const int l15 = lane & 15, kq4 = (lane >> 4) * 4, row = tid >> 4, col = tid & 15;
    const unsigned shard = xcc_id();
    unsigned* const status = a.status;
    Ctx c = {tid, lane, wave, wg, G, nrb, B, row, col, lds + L_red, 0, status, reinterpret_cast<int*>(lds), nullptr};
    float* const st = lds + L_fixed;                            // per job: carry (d gh_{t+1} . W_hh) | direct term z * d of step t+1
    const long long Bp = 16ll * nrb, BH = (long long)B * H;
    const int tU = wg % TILES, rb0 = wg / TILES, drb = G / TILES;
    const int nj = wg < TILES * nrb ? (TILES * nrb - wg + G - 1) / G : 0;
    const int unit = 16 * tU + col;
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.w, 3ll * H * H * 4);
    const unsigned kw = (unsigned)(wave * (3 * H / 4) + kq4) * 4u;
    for (int n = 0; n < nj; ++n) { st[n * 512 + tid] = 0.f; st[n * 512 + 256 + tid] = 0.f; }
    auto nopre = [](int, int) { return PreNoneG{0.f}; };
    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        // d gh is handed on through one of two buffers (by step parity): a buffer is rewritten two steps later, when every workgroup has
        // passed the seam of the step in between -- which it reaches only after its carry jobs have read this one
        const __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.Dfm + (long long)(s & 1) * Bp * 3 * H, Bp * 3 * H * 4);
        // cell backward, element-wise, by the owners
        for (int n = 0; n < nj; ++n) {
            const int rb = rb0 + n * drb, grow = rb * 16 + row, gr = min(grow, B - 1);
            const float* sp = a.save + ((long long)t * B + gr) * 4 * H;
            const float r = sp[unit], z = sp[H + unit], nn = sp[2 * H + unit], ghn = sp[3 * H + unit];
            const float d = a.dout[((long long)gr * T + t) * H + unit] + st[n * 512 + 256 + tid] + st[n * 512 + tid];
            const float dn = d * (1.f - z);
            const float dz = d * (a.hall[(long long)t * BH + (long long)gr * H + unit] - nn);
            const float dpn = dn * (1.f - nn * nn);
            const float dpr = dpn * ghn * r * (1.f - r);
            const float dpz = dz * z * (1.f - z);
            st[n * 512 + 256 + tid] = d * z;
            const float dg[3] = {dpr, dpz, dpn * r};
            if (grow < B) {
                float* gi_o = a.dgi + ((long long)grow * T + t) * 3 * H;
                float* gh_o = a.dgh + ((long long)t * B + grow) * 3 * H;
                gi_o[unit] = dpr; gi_o[H + unit] = dpz; gi_o[2 * H + unit] = dpn;
                gh_o[unit] = dg[0]; gh_o[H + unit] = dg[1]; gh_o[2 * H + unit] = dg[2];
            }
            if (t > 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const f32x4 q = gather4(dg[g]);
                    if ((col & 3) == 0) bstore4_sc1(rsD, fm_chunk(rb, g * TG + tU, KB) + (unsigned)(((col >> 2) * 16 + row) * 16), q);
                }
            }
        }
        if (t == 0) break;
        seam_arrive(a.sync, shard);
        // carry of step t-1: d gh_t . W_hh   (K = 3H; the weight rows of the owned units, k-contiguous in the transposed copy)
        if (!run_mt<NB, 1, PreNoneG>(
                c, TILES, [&](int, int i) { return bload4<0>(rsW, (unsigned)((16 * tU + l15) * 3 * H) * 4u + kw + 64u * i); }, rsD, KB, a.sync,
                (unsigned)G * (unsigned)(s + 1), 0x310u, -1, nopre,
                [&](const float (&v)[1], const PreNoneG&, int, int n) { st[n * 512 + tid] = v[0]; }))
            return;
    }
}

constexpr int GH = 1024;

struct GruPlanP { int G, nrb, jobs; size_t lds_bytes, o_hfm, o_wt, o_dfm, total; };

bool gru_persistent_plan(int B, int T, int H, bool backward, GruPlanP& pl) {
    if (H != GH || B < 1 || T < 1) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const int G = cus >= 256 ? 256 : (cus >= 128 ? 128 : (cus >= 64 ? 64 : 0));
    if (!G) return false;
    pl.G = G;
    pl.nrb = (B + 15) / 16;
    pl.jobs = ((GH / 16) * pl.nrb + G - 1) / G;
    if (pl.jobs > MAXJ * 2) return false;
    pl.lds_bytes = (size_t)(4 + 2 * 5120 + pl.jobs * (backward ? 512 : 256)) * sizeof(float);
    if (!grid_fits(backward ? reinterpret_cast<const void*>(gru_seq_bwd_persistent_kernel<GH>) : reinterpret_cast<const void*>(gru_seq_fwd_persistent_kernel<GH>),
                   G, cus, pl.lds_bytes)) return false;
    const size_t Bp = 16 * (size_t)pl.nrb, f = sizeof(float);
    size_t o = al256((size_t)SYNC_WORDS * 4);
    pl.o_hfm = o; if (!backward) o += al256(2 * Bp * GH * f);
    pl.o_wt = o; if (backward) o += al256((size_t)GH * 3 * GH * f);
    pl.o_dfm = o; if (backward) o += al256(2 * Bp * 3 * GH * f);
    pl.total = o;
    return true;
}

template <class K>
int launch_gru(K kern, const PGru& a, const GruPlanP& pl, hipStream_t st, const char* what) {
    if (func_attr_once(reinterpret_cast<const void*>(kern))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            set_error("%s: cannot raise the dynamic LDS limit", what);
            return EKAID_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(pl.G), dim3(256), pl.lds_bytes, st, a);
    return check_launch(what);
}

}  // namespace
}  // namespace ekaid

using namespace ekaid;

extern "C" size_t ekaid_gru_seq_persistent_workspace_bytes(int B, int T, int H, int backward) {
    GruPlanP pl;
    return gru_persistent_plan(B, T, H, backward != 0, pl) ? pl.total : 0;
}

extern "C" int ekaid_gru_seq_fwd_persistent(const float* gi, const float* w_hh, const float* b_hh, int B, int T, int H, float* hall, float* save,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    GruPlanP pl;
    if (!gi || !w_hh || !b_hh || !hall || !save) { set_error("ekaid_gru_seq_fwd_persistent: null pointer"); return EKAID_ERR_ARG; }
    if (!gru_persistent_plan(B, T, H, false, pl)) { set_error("ekaid_gru_seq_fwd_persistent: unsupported shape (H=%d B=%d) or device", H, B); return EKAID_ERR_UNSUPPORTED; }
    if (!workspace || workspace_bytes < pl.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        set_error("ekaid_gru_seq_fwd_persistent: workspace too small or misaligned (%zu < %zu)", workspace_bytes, pl.total);
        return EKAID_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    PGru a;
    memset(&a, 0, sizeof(a));
    a.T = T; a.B = B; a.G = pl.G; a.nrb = pl.nrb; a.jobs = pl.jobs;
    a.gi = gi; a.w = w_hh; a.b_hh = b_hh; a.hall = hall; a.save = save;
    a.Hfm = (float*)((char*)workspace + pl.o_hfm);
    a.sync = (unsigned*)workspace;
    a.status = seq_status_ptr();
    if (!a.status) { set_error("persistent sequence kernel: cannot allocate the status word"); return EKAID_ERR_LAUNCH; }
    if (hipMemsetAsync(workspace, 0, al256((size_t)SYNC_WORDS * 4), st) != hipSuccess) { (void)hipGetLastError(); set_error("ekaid_gru_seq_fwd_persistent: memset failed"); return EKAID_ERR_LAUNCH; }
    count_launch(EKAID_COUNT_GRU_FWD_PERSISTENT);
    return launch_gru(gru_seq_fwd_persistent_kernel<GH>, a, pl, st, "gru_seq_fwd_persistent");
}

extern "C" int ekaid_gru_seq_bwd_persistent(const float* dout, const float* w_hh, const float* hall, const float* save, int B, int T, int H,
                                            float* dgi, float* dgh, void* workspace, size_t workspace_bytes, void* stream) {
    GruPlanP pl;
    if (!dout || !w_hh || !hall || !save || !dgi || !dgh) { set_error("ekaid_gru_seq_bwd_persistent: null pointer"); return EKAID_ERR_ARG; }
    if (!gru_persistent_plan(B, T, H, true, pl)) { set_error("ekaid_gru_seq_bwd_persistent: unsupported shape (H=%d B=%d) or device", H, B); return EKAID_ERR_UNSUPPORTED; }
    if (!workspace || workspace_bytes < pl.total || (reinterpret_cast<uintptr_t>(workspace) & 255)) {
        set_error("ekaid_gru_seq_bwd_persistent: workspace too small or misaligned (%zu < %zu)", workspace_bytes, pl.total);
        return EKAID_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PGru a;
    memset(&a, 0, sizeof(a));
    a.T = T; a.B = B; a.G = pl.G; a.nrb = pl.nrb; a.jobs = pl.jobs;
    a.w = (const float*)(ws + pl.o_wt); a.hall = const_cast<float*>(hall); a.save = const_cast<float*>(save);
    a.dout = dout; a.dgi = dgi; a.dgh = dgh;
    a.Dfm = (float*)(ws + pl.o_dfm);
    a.sync = (unsigned*)workspace;
    a.status = seq_status_ptr();
    if (!a.status) { set_error("persistent sequence kernel: cannot allocate the status word"); return EKAID_ERR_LAUNCH; }
    const int rc = ekaid_transpose(w_hh, 3 * GH, GH, GH, (float*)(ws + pl.o_wt), 3 * GH, stream);      // W_hh^T [H][3H]
    if (rc) return rc;
    if (hipMemsetAsync(workspace, 0, al256((size_t)SYNC_WORDS * 4), st) != hipSuccess) { (void)hipGetLastError(); set_error("ekaid_gru_seq_bwd_persistent: memset failed"); return EKAID_ERR_LAUNCH; }
    count_launch(EKAID_COUNT_GRU_BWD_PERSISTENT);
    return launch_gru(gru_seq_bwd_persistent_kernel<GH>, a, pl, st, "gru_seq_bwd_persistent");
}
