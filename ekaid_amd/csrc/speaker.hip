// Fused per-step kernels of the answer decoder (DynamicCore.forward, dynamic_speaker_change_pos.py:94-131).
//
// One decoder step at batch 256 is a chain of ~25 dependent launches of a few microseconds each; the chains below are
// row-local (every batch row is independent), so each becomes ONE kernel with one 256-thread workgroup per batch row:
//   module-weight chain : fc(h_att) -> softmax over the 3 modules -> weighted mix of (bef, diff, aft)   (:104-113)
//   POS chain           : relu(pos1(h_lang)+b) -> dropout -> weight_pos -> dropout -> softmax -> pos2   (:115-121)
//   gate derivative     : d(gate) through sigmoid for gated = gate * att                                 (:122-125)
// and their backward counterparts.  All matrices involved are tiny (3 x R, 16 x R, R x 16): they are read from L2
// by every workgroup; the batch-row vectors are staged through LDS.
#include "common.h"
#include "../../include/ekaid_hip.h"

namespace ekaid {

#define ST ((hipStream_t)stream)
#define INVK(p) ((p) > 0.f ? 1.f / (1.f - (p)) : 1.f)
constexpr int SPK_MAXR = 2048;      // staged row length (rnn_size / att dim) limit of the LDS buffers below
constexpr int SPK_MAXP = 64;        // POS classes limit

// block-wide sum of NV values per thread (256 threads); result valid in every thread
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [NV][4] */) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                // protect `red` from the previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 4 + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (red[i * 4] + red[i * 4 + 1]) + (red[i * 4 + 2] + red[i * 4 + 3]);
}

// ---- module weights + attended feature --------------------------------------------------------------------------
// mw = softmax(h . w_fc^T + b_fc)  (3 modules);  att = mw0*bef + mw1*diff + mw2*aft, written twice: contiguous `att`
// and (optionally) a strided copy `att2` (the [ppos | att] operand of the gate MLP GEMM)
__global__ __launch_bounds__(256) void spk_modmix_fwd_kernel(const float* h, int R, const float* w_fc, const float* b_fc,
                                                             const float* bef, const float* diff, const float* aft, int A,
                                                             float* mw, float* att, float* att2, int ld2) {
    __shared__ float red[12];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* hb = h + (long long)b * R;
    float z[3] = {0.f, 0.f, 0.f};
    for (int k = tid; k < R; k += 256) {
        const float x = hb[k];
        z[0] += x * w_fc[k]; z[1] += x * w_fc[R + k]; z[2] += x * w_fc[2 * R + k];
    }
    block_sum<3>(z, red);
    z[0] += b_fc[0]; z[1] += b_fc[1]; z[2] += b_fc[2];
    const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
    const float e0 = __expf(z[0] - mx), e1 = __expf(z[1] - mx), e2 = __expf(z[2] - mx);
    const float inv = 1.f / (e0 + e1 + e2);
    const float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
    if (tid == 0) { mw[b * 3] = w0; mw[b * 3 + 1] = w1; mw[b * 3 + 2] = w2; }
    for (int k = tid; k < A; k += 256) {
        const long long i = (long long)b * A + k;
        const float v = w0 * bef[i] + w1 * diff[i] + w2 * aft[i];
        att[i] = v;
        if (att2) att2[(long long)b * ld2 + k] = v;
    }
}

// d_att = d_att_a (+ d_gated * gate when d_gated != NULL);  d{bef,diff,aft} += mw_c * d_att;  dmw_c = <d_att, feat_c>;
// dz = softmax'(mw) dmw -> dz_out;  dh += dz . w_fc
__global__ __launch_bounds__(256) void spk_modmix_bwd_kernel(const float* mw, const float* bef, const float* diff, const float* aft,
                                                             int A, const float* d_att_a, int lda, const float* d_gated,
                                                             const float* gate, const float* w_fc, int R, float* dz_out,
                                                             float* d_bef, float* d_diff, float* d_aft, float* dh) {
    __shared__ float red[12];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float w0 = mw[b * 3], w1 = mw[b * 3 + 1], w2 = mw[b * 3 + 2];
    float s[3] = {0.f, 0.f, 0.f};
    for (int k = tid; k < A; k += 256) {
        const long long i = (long long)b * A + k;
        float d = d_att_a[(long long)b * lda + k];
        if (d_gated) d += d_gated[i] * gate[i];
        s[0] += d * bef[i]; s[1] += d * diff[i]; s[2] += d * aft[i];
        d_bef[i] += w0 * d; d_diff[i] += w1 * d; d_aft[i] += w2 * d;
    }
    block_sum<3>(s, red);
    const float dot = w0 * s[0] + w1 * s[1] + w2 * s[2];
    const float dz0 = w0 * (s[0] - dot), dz1 = w1 * (s[1] - dot), dz2 = w2 * (s[2] - dot);
    if (tid == 0) { dz_out[b * 3] = dz0; dz_out[b * 3 + 1] = dz1; dz_out[b * 3 + 2] = dz2; }
    for (int k = tid; k < R; k += 256)
        dh[(long long)b * R + k] += dz0 * w_fc[k] + dz1 * w_fc[R + k] + dz2 * w_fc[2 * R + k];
}

// ---- POS chain ---------------------------------------------------------------------------------------------------
// v = dropout_pv(relu(x + b1)) (R)  ->  dpos = dropout_p5(w_wp v + b_wp) (P)  ->  pw = softmax(dpos)  ->  pp = w_p2 pw + b_p2 (R)
// dropout masks: keep(seed_v, b*R + k) and keep(seed_d, b*P + j) -- the conventions of ekaid_bias_act_drop and of the
// GEMM epilogue dropout, which this kernel replaces
__global__ __launch_bounds__(256) void spk_pos_fwd_kernel(const float* x, int ldx, const float* b1, int R, const float* w_wp,
                                                          const float* b_wp, int P, const float* w_p2, const float* b_p2,
                                                          float pv, float invv, DropSeed seed_v, float p5, float inv5,
                                                          DropSeed seed_d, float* vp, float* dp, float* pw, float* pp, int ldpp) {
    __shared__ float sv[SPK_MAXR];
    __shared__ float sz[SPK_MAXP];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < R; k += 256) {
        float v = x[(long long)b * ldx + k] + b1[k];
        v = (v > 0.f ? v : 0.f) * dropout_scale(seed_v, (uint64_t)b * R + k, pv, invv);
        sv[k] = v;
        vp[(long long)b * R + k] = v;
    }
    __syncthreads();
    // P dot products of length R: 16 lanes per output, outputs processed 16 at a time
    const int l = tid & 15, jg = tid >> 4;
    for (int j0 = 0; j0 < P; j0 += 16) {
        const int j = j0 + jg;
        float s = 0.f;
        if (j < P) {
            const float* wr = w_wp + (long long)j * R;
            for (int k = l; k < R; k += 16) s += wr[k] * sv[k];
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
        if (j < P && l == 0) {
            const float zz = (s + b_wp[j]) * dropout_scale(seed_d, (uint64_t)b * P + j, p5, inv5);
            sz[j] = zz;
            dp[(long long)b * P + j] = zz;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = sz[0];
        for (int j = 1; j < P; ++j) mx = fmaxf(mx, sz[j]);
        float sum = 0.f;
        for (int j = 0; j < P; ++j) { const float e = __expf(sz[j] - mx); sz[j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < P; ++j) { sz[j] *= inv; pw[(long long)b * P + j] = sz[j]; }
    }
    __syncthreads();
    for (int k = tid; k < R; k += 256) {
        const float* wr = w_p2 + (long long)k * P;
        float s = b_p2[k];
        for (int j = 0; j < P; ++j) s += wr[j] * sz[j];
        pp[(long long)b * ldpp + k] = s;
    }
}

// backward of the chain: dpp (R, leading dim lddpp) -> dpw = w_p2^T dpp -> softmax' -> + dhead (gradient arriving at dpos
// through the log-softmax head) -> dropout_p5 -> dwp_out (P);  dv = w_wp^T dwp -> dropout_pv, relu' (from the saved
// output vp) -> dx_out (R, leading dim lddx)
__global__ __launch_bounds__(256) void spk_pos_bwd_kernel(const float* dpp, int lddpp, const float* pw, const float* dhead, int P,
                                                          const float* w_p2, const float* w_wp, const float* vp, int R, float pv,
                                                          float invv, DropSeed seed_v, float p5, float inv5, DropSeed seed_d,
                                                          float* dwp_out, float* dx_out, int lddx) {
    __shared__ float sd[SPK_MAXR];
    __shared__ float sg[SPK_MAXP];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < R; k += 256) sd[k] = dpp[(long long)b * lddpp + k];
    __syncthreads();
    // dpw_j = sum_k dpp[k] * w_p2[k][j]: thread (kq, j) walks rows k = kq, kq + 256/PG, .. so that a wave reads whole
    // 64-byte rows of w_p2 (P = 16: 4 rows per 16 lanes... coalesced), partial sums combined through LDS
    __shared__ float sp[256];
    {
        const int PG = P <= 16 ? 16 : (P <= 32 ? 32 : 64);        // lanes per row of w_p2
        const int j = tid % PG, kq = tid / PG, KQ = 256 / PG;
        float s = 0.f;
        if (j < P)
            for (int k = kq; k < R; k += KQ) s += sd[k] * w_p2[(long long)k * P + j];
        sp[tid] = s;
        __syncthreads();
        if (tid < P) {
            float t = 0.f;
            for (int q = 0; q < KQ; ++q) t += sp[q * PG + tid];
            sg[tid] = t;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float dot = 0.f;
        for (int j = 0; j < P; ++j) dot += pw[(long long)b * P + j] * sg[j];
        for (int j = 0; j < P; ++j) {
            float g = pw[(long long)b * P + j] * (sg[j] - dot);
            if (dhead) g += dhead[(long long)b * P + j];
            g *= dropout_scale(seed_d, (uint64_t)b * P + j, p5, inv5);
            sg[j] = g;
            dwp_out[(long long)b * P + j] = g;
        }
    }
    __syncthreads();
    for (int k = tid; k < R; k += 256) {
        float s = 0.f;
        for (int j = 0; j < P; ++j) s += sg[j] * w_wp[(long long)j * R + k];
        const float y = vp[(long long)b * R + k];
        dx_out[(long long)b * lddx + k] = s * dropout_scale(seed_v, (uint64_t)b * R + k, pv, invv) * (y > 0.f ? 1.f : 0.f);
    }
}

// gated = gate * att, gate = sigmoid(z):  dz = d_gated * att * gate * (1 - gate)
__global__ __launch_bounds__(256) void spk_gate_bwd_kernel(const float* d_gated, const float* att, const float* gate, long long n,
                                                           float* dz) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = gate[i];
        dz[i] = d_gated[i] * att[i] * g * (1.f - g);
    }
}

// ---- scheduled sampling (dynamic_speaker_change_pos.py:197-208): with probability ss_prob per sample the teacher token of this
// step is replaced by a draw from the previous step's output distribution exp(logp_prev).  One thread per sample: uniform u1
// decides, inverse-CDF walk over the V <= a few hundred classes draws (the reference uses torch.multinomial: same
// distribution, different random stream -- as for dropout, streams cannot match).
__global__ __launch_bounds__(64) void ss_tokens_kernel(const float* __restrict__ logp_prev, const long long* __restrict__ teacher,
                                                       int ld_teacher, int B, int V, float ss_prob, DropSeed seed,
                                                       long long* __restrict__ it) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    long long tok = teacher[(long long)b * ld_teacher];
    const uint64_t sd = drop_seed_value(seed);
    const float u1 = (float)(hash_u32(sd, (uint64_t)b) >> 8) * (1.0f / 16777216.0f);
    if (u1 < ss_prob) {
        const float u2 = (float)(hash_u32(sd + 0x5851F42D4C957F2Dull, (uint64_t)b) >> 8) * (1.0f / 16777216.0f);
        const float* p = logp_prev + (long long)b * V;
        float tot = 0.f;
        for (int v = 0; v < V; ++v) tot += __expf(p[v]);
        const float target = u2 * tot;
        float run = 0.f;
        int pick = V - 1;
        for (int v = 0; v < V; ++v) {
            run += __expf(p[v]);
            if (run > target) { pick = v; break; }
        }
        tok = pick;
    }
    it[b] = tok;
}

}  // namespace ekaid

using namespace ekaid;

extern "C" int ekaid_ss_tokens(const float* logp_prev, const int64_t* teacher, int ld_teacher, int B, int V, float ss_prob,
                               uint64_t seed, int64_t* it, void* stream) {
    if (!logp_prev || !teacher || !it || V < 1) { set_error("ss_tokens: null pointer / empty vocabulary"); return EKAID_ERR_ARG; }
    if (B <= 0) return EKAID_OK;
    hipLaunchKernelGGL(ss_tokens_kernel, dim3((B + 63) / 64), dim3(64), 0, ST, logp_prev, (const long long*)teacher, ld_teacher, B, V, ss_prob,
                       DS(seed), (long long*)it);
    return check_launch("ss_tokens");
}

extern "C" int ekaid_spk_modmix_fwd(const float* h, int B, int R, const float* w_fc, const float* b_fc, const float* bef,
                                    const float* diff, const float* aft, int A, float* mw, float* att, float* att2, int ld2,
                                    void* stream) {
    if (B <= 0) return EKAID_OK;
    hipLaunchKernelGGL(spk_modmix_fwd_kernel, dim3(B), dim3(256), 0, ST, h, R, w_fc, b_fc, bef, diff, aft, A, mw, att, att2, ld2);
    return check_launch("spk_modmix_fwd");
}

extern "C" int ekaid_spk_modmix_bwd(const float* mw, const float* bef, const float* diff, const float* aft, int B, int A,
                                    const float* d_att_a, int lda, const float* d_gated, const float* gate, const float* w_fc,
                                    int R, float* dz_out, float* d_bef, float* d_diff, float* d_aft, float* dh, void* stream) {
    if (B <= 0) return EKAID_OK;
    hipLaunchKernelGGL(spk_modmix_bwd_kernel, dim3(B), dim3(256), 0, ST, mw, bef, diff, aft, A, d_att_a, lda, d_gated, gate, w_fc, R,
                       dz_out, d_bef, d_diff, d_aft, dh);
    return check_launch("spk_modmix_bwd");
}

extern "C" int ekaid_spk_pos_fwd(const float* x, int ldx, const float* b1, int B, int R, const float* w_wp, const float* b_wp, int P,
                                 const float* w_p2, const float* b_p2, float p_v, uint64_t seed_v, float p_d, uint64_t seed_d,
                                 float* vp, float* dp, float* pw, float* pp, int ldpp, void* stream) {
    if (R > SPK_MAXR || P > SPK_MAXP) { set_error("ekaid_spk_pos_fwd: R <= %d and P <= %d required", SPK_MAXR, SPK_MAXP); return EKAID_ERR_UNSUPPORTED; }
    if (B <= 0) return EKAID_OK;
    hipLaunchKernelGGL(spk_pos_fwd_kernel, dim3(B), dim3(256), 0, ST, x, ldx, b1, R, w_wp, b_wp, P, w_p2, b_p2, p_v, INVK(p_v), DS(seed_v),
                       p_d, INVK(p_d), DS(seed_d), vp, dp, pw, pp, ldpp);
    return check_launch("spk_pos_fwd");
}

extern "C" int ekaid_spk_pos_bwd(const float* dpp, int lddpp, const float* pw, const float* dhead, int B, int P, const float* w_p2,
                                 const float* w_wp, const float* vp, int R, float p_v, uint64_t seed_v, float p_d, uint64_t seed_d,
                                 float* dwp_out, float* dx_out, int lddx, void* stream) {
    if (R > SPK_MAXR || P > SPK_MAXP) { set_error("ekaid_spk_pos_bwd: R <= %d and P <= %d required", SPK_MAXR, SPK_MAXP); return EKAID_ERR_UNSUPPORTED; }
    if (B <= 0) return EKAID_OK;
    hipLaunchKernelGGL(spk_pos_bwd_kernel, dim3(B), dim3(256), 0, ST, dpp, lddpp, pw, dhead, P, w_p2, w_wp, vp, R, p_v, INVK(p_v),
                       DS(seed_v), p_d, INVK(p_d), DS(seed_d), dwp_out, dx_out, lddx);
    return check_launch("spk_pos_bwd");
}

extern "C" int ekaid_spk_gate_bwd(const float* d_gated, const float* att, const float* gate, long long n, float* dz, void* stream) {
    if (n <= 0) return EKAID_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(spk_gate_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, d_gated, att, gate, n, dz);
    return check_launch("spk_gate_bwd");
}
