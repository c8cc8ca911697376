// timing harness for the deterministic embedding backward (scan only / scan + gather)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(64) void k(const int64_t* tok, long long ntok, const float* dout, int E, int ld_out, float* dtable) {
    constexpr int U = 16;
    const int64_t id = blockIdx.x;
    const int lane = threadIdx.x;
    const int e = blockIdx.y * 64 + lane;
    const bool col = e < E;
    const float* dcol = dout + (col ? e : 0);
    float acc = 0.f;
    for (long long sbase = 0; sbase < ntok; sbase += 64 * U) {
        bool hit[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long t = sbase + 64 * u + lane;
            hit[u] = t < ntok && tok[t] == id;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned long long m = __ballot(hit[u]);
            if (MODE == 0) { acc += (float)__popcll(m); continue; }
            const float* rowbase = dcol + (sbase + 64 * u) * ld_out;
            while (m) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (m) { const int b = __builtin_ctzll(m); m &= m - 1; v[q] = rowbase[(long long)b * ld_out]; }
                }
                acc += v[0]; acc += v[1]; acc += v[2]; acc += v[3];
            }
        }
    }
    if (col) dtable[id * E + e] += acc;
}
template <int MODE>
static float run(const int64_t* tok, int ntok, const float* d, int E, int V, float* dt) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(V, (E + 63) / 64), dim3(64), 0, 0, tok, (long long)ntok, d, E, E, dt);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(V, (E + 63) / 64), dim3(64), 0, 0, tok, (long long)ntok, d, E, E, dt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.f;
}
int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int ntok = cfg == 2 ? 23040 : (cfg == 1 ? 4096 : 5120), V = 148, E = cfg == 0 ? 300 : 300;
        std::vector<int64_t> h(ntok);
        for (int i = 0; i < ntok; ++i) h[i] = rand() % V;
        int64_t* tok; float *d, *dt;
        hipMalloc(&tok, ntok * 8); hipMalloc(&d, (size_t)ntok * E * 4); hipMalloc(&dt, V * E * 4);
        hipMemcpy(tok, h.data(), ntok * 8, hipMemcpyHostToDevice);
        hipMemset(d, 0, (size_t)ntok * E * 4); hipMemset(dt, 0, V * E * 4);
        printf("ntok %d V %d E %d: scan only %.1f us | scan + gather %.1f us\n", ntok, V, E, run<0>(tok, ntok, d, E, V, dt), run<1>(tok, ntok, d, E, V, dt));
    }
    return 0;
}
