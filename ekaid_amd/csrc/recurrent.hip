// Point-wise halves of the recurrent cells (the gate pre-activations come from ekaid_gemm_f32).
//   GRU  (question encoder, ref: language_model.py:66-70,113; torch gate order r,z,n, b_hn inside r*(.))
//   LSTM (speaker's two LSTMCells, ref: dynamic_speaker_change_pos.py:57,88,102,126; gate order i,f,g,o)
#include "common.h"
#include "../../include/ekaid_hip.h"

namespace ekaid {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* gi, int ldgi, const float* gh, int ldgh, const float* h,
                                                           int ldh, int B, int H, float* hnew, int ldhn, float* save) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / H;
        const int j = (int)(i % H);
        const float* gip = gi + b * ldgi;
        const float* ghp = gh + b * ldgh;
        const float r = sigm(gip[j] + ghp[j]);
        const float z = sigm(gip[H + j] + ghp[H + j]);
        const float ghn = ghp[2 * H + j];
        const float nn = tanhf(gip[2 * H + j] + r * ghn);
        hnew[b * ldhn + j] = (1.f - z) * nn + z * h[b * ldh + j];
        float* sp = save + b * 4 * H;
        sp[j] = r; sp[H + j] = z; sp[2 * H + j] = nn; sp[3 * H + j] = ghn;
    }
}
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* dhnew, int lddh, const float* dh_carry, const float* save,
                                                           const float* h, int ldh, int B, int H, float* dgi, int lddgi, float* dgh,
                                                           float* dh) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / H;
        const int j = (int)(i % H);
        const float* sp = save + b * 4 * H;
        const float r = sp[j], z = sp[H + j], nn = sp[2 * H + j], ghn = sp[3 * H + j];
        const float d = (dhnew ? dhnew[b * lddh + j] : 0.f) + (dh_carry ? dh_carry[b * H + j] : 0.f);
        const float dn = d * (1.f - z);
        const float dz = d * (h[b * ldh + j] - nn);
        const float dpn = dn * (1.f - nn * nn);
        const float dpr = dpn * ghn * r * (1.f - r);
        const float dpz = dz * z * (1.f - z);
        float* gi_o = dgi + b * lddgi;
        float* gh_o = dgh + b * 3 * H;
        gi_o[j] = dpr; gi_o[H + j] = dpz; gi_o[2 * H + j] = dpn;
        gh_o[j] = dpr; gh_o[H + j] = dpz; gh_o[2 * H + j] = dpn * r;
        dh[b * H + j] = d * z;
    }
}
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* gx, const float* gh, int ldgh, const float* c, int B, int H,
                                                            float* h2, float* c2, float* save) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / H;
        const int j = (int)(i % H);
        const float* x = gx + b * 4 * H;
        const float* y = gh + b * ldgh;
        const float ig = sigm(x[j] + y[j]);
        const float fg = sigm(x[H + j] + y[H + j]);
        const float gg = tanhf(x[2 * H + j] + y[2 * H + j]);
        const float og = sigm(x[3 * H + j] + y[3 * H + j]);
        const float cn = fg * c[i] + ig * gg;
        c2[i] = cn;
        h2[i] = og * tanhf(cn);
        float* sp = save + b * 4 * H;
        sp[j] = ig; sp[H + j] = fg; sp[2 * H + j] = gg; sp[3 * H + j] = og;
    }
}
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* dh2, const float* dh2b, const float* dc2, const float* save,
                                                            const float* c, const float* c2, int B, int H, float* dgates, int lddg,
                                                            float* dc) {
    const long long n = (long long)B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / H;
        const int j = (int)(i % H);
        const float* sp = save + b * 4 * H;
        const float ig = sp[j], fg = sp[H + j], gg = sp[2 * H + j], og = sp[3 * H + j];
        const float tc = tanhf(c2[i]);
        const float dh = (dh2 ? dh2[i] : 0.f) + (dh2b ? dh2b[i] : 0.f);
        const float dct = (dc2 ? dc2[i] : 0.f) + dh * og * (1.f - tc * tc);
        float* o = dgates + b * lddg;
        o[j] = dct * gg * ig * (1.f - ig);
        o[H + j] = dct * c[i] * fg * (1.f - fg);
        o[2 * H + j] = dct * ig * (1.f - gg * gg);
        o[3 * H + j] = dh * tc * og * (1.f - og);
        dc[i] = dct * fg;
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
#define ST ((hipStream_t)stream)

extern "C" int ekaid_gru_cell_fwd(const float* gi, int ldgi, const float* gh, int ldgh, const float* h, int ldh, int B, int H,
                                  float* hnew, int ldhn, float* save, void* stream) {
    hipLaunchKernelGGL(gru_cell_fwd_kernel, g1((long long)B * H), dim3(256), 0, ST, gi, ldgi, gh, ldgh, h, ldh, B, H, hnew, ldhn, save);
    return check_launch("gru_cell_fwd");
}
extern "C" int ekaid_gru_cell_bwd(const float* dhnew, int lddh, const float* dh_carry, const float* save, const float* h, int ldh,
                                  int B, int H, float* dgi, int lddgi, float* dgh, float* dh, void* stream) {
    hipLaunchKernelGGL(gru_cell_bwd_kernel, g1((long long)B * H), dim3(256), 0, ST, dhnew, lddh, dh_carry, save, h, ldh, B, H, dgi, lddgi,
                       dgh, dh);
    return check_launch("gru_cell_bwd");
}
extern "C" int ekaid_lstm_cell_fwd(const float* gx, const float* gh, int ldgh, const float* c, int B, int H, float* h2, float* c2,
                                   float* save, void* stream) {
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, g1((long long)B * H), dim3(256), 0, ST, gx, gh, ldgh, c, B, H, h2, c2, save);
    return check_launch("lstm_cell_fwd");
}
extern "C" int ekaid_lstm_cell_bwd(const float* dh2, const float* dh2b, const float* dc2, const float* save, const float* c,
                                   const float* c2, int B, int H, float* dgates, int lddg, float* dc, void* stream) {
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, g1((long long)B * H), dim3(256), 0, ST, dh2, dh2b, dc2, save, c, c2, B, H, dgates, lddg,
                       dc);
    return check_launch("lstm_cell_bwd");
}
