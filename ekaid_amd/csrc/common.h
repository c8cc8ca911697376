// Shared device helpers for the ekaid HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EKAID_OK 0
#define EKAID_ERR_ARG (-1)
#define EKAID_ERR_LAUNCH (-2)
#define EKAID_ERR_UNSUPPORTED (-3)

namespace ekaid {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// per-device state (api.hip): created by ekaid_init(device) or on first use of a device; no other mutable globals
struct DeviceState;
DeviceState& device_state();                       // of the calling thread's current HIP device
int gemm_mode();                                   // 0 = f32 MFMA, 1 = exact bf16 split (per device)
int set_gemm_mode(int mode);
int seq_mode();                                    // 0 = step-wise sequence chains, 1 = persistent sequence kernels where supported (per device)
#ifndef EKAID_GEMM_MODE_DEFAULT
#define EKAID_GEMM_MODE_DEFAULT 2                  // the two-term f16 split for products that carry operand bounds, the exact 3-way bf16 split for the rest
#endif
#ifndef EKAID_SEQ_MODE_DEFAULT
#define EKAID_SEQ_MODE_DEFAULT 1
#endif
const unsigned long long* dropout_epoch_ptr();     // device word folded into every dropout seed, or NULL
constexpr size_t EKAID_SCRATCH_BYTES = 2 * 2048 * sizeof(double) + (size_t(1) << 20);    // reduction slots + column-sum partials (pointwise.hip)
double* device_scratch(size_t bytes);              // per-device reduction scratch (allocated at first use / by ekaid_init), or NULL
unsigned* seq_status_ptr();                        // per-device word set (and left set) by a persistent sequence kernel whose seam timed out
void count_launch(int which);                      // bumps a launch counter (ekaid_launch_count, include/ekaid_hip.h)
bool func_attr_once(const void* kernel);           // true exactly once per (device, kernel function): "raise the LDS limit" bookkeeping

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_TANH: return 1.f - y * y;
        case ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ---- counter-based dropout RNG -------------------------------------------------
// keep(i) is a pure function of (seed, i): the backward pass regenerates the mask,
// nothing is stored.  murmur3 finaliser over a 64-bit counter folded to 32 bits.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33;
    return (uint32_t)x;
}
// returns the multiplier applied to an element: 0 (dropped) or 1/(1-p) (kept)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    float u = (float)(hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);   // [0,1)
    return u < p ? 0.f : inv_keep;
}
// A dropout site's seed as the kernels see it: the caller's 64-bit seed plus an optional DEVICE-resident epoch word
// (ekaid_set_dropout_epoch_ptr).  A captured hipGraph freezes every by-value seed; the epoch word is advanced by a
// kernel inside the graph (ekaid_dropout_epoch_advance), so every replay draws fresh masks while the forward and the
// backward launches of one replay still agree.  epoch pointer NULL (the default) = the seed alone.
struct DropSeed {
    uint64_t s;
    const unsigned long long* ep;
};
__device__ __forceinline__ uint64_t drop_seed_value(const DropSeed& d) {
    return d.ep ? d.s + (uint64_t)d.ep[0] * 0x9E3779B97F4A7C15ull : d.s;
}
__device__ __forceinline__ float dropout_scale(const DropSeed& d, uint64_t idx, float p, float inv_keep) {
    if (p <= 0.f) return 1.f;
    return dropout_scale(drop_seed_value(d), idx, p, inv_keep);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace ekaid
#define DS(seed) (ekaid::DropSeed{(uint64_t)(seed), ekaid::dropout_epoch_ptr()})
namespace ekaid {

}  // namespace ekaid
