// Error bookkeeping + version for libekaid_hip.so.
#include "common.h"
#include "../../include/ekaid_hip.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace ekaid {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return EKAID_ERR_LAUNCH;
    }
    return EKAID_OK;
}

// ---- per-device state: the only mutable state of the library besides the thread-local error string.  Every field is
// an atomic (the list of kernels whose LDS limit has been raised: behind a mutex), so calls from different host threads / on different
// streams never race on it.
constexpr int MAX_DEVICES = 64;
struct DeviceState {
    std::atomic<int> ready{0};
    std::atomic<int> gemm_mode{-1};
    std::atomic<int> seq_mode{-1};
    std::atomic<const unsigned long long*> epoch{nullptr};
    std::atomic<double*> scratch{nullptr};
    std::atomic<unsigned*> seq_status{nullptr};   // device word the persistent sequence kernels set when a seam times out (sticky)        // two-stage reduction partials (pointwise.hip), device memory of THIS device
    std::mutex attr_mu;
    std::vector<const void*> attr_done;           // kernel functions whose dynamic-LDS limit has been raised on this device
};
static DeviceState g_dev[MAX_DEVICES];

static int default_gemm_mode() {
    const char* m = getenv("EKAID_GEMM_MODE");       // "f32" / "0": f32 MFMA; "f16x2" / "2": two-term f16 split where operand bounds are given; default: exact split
    if (m && (strcmp(m, "2") == 0 || strcmp(m, "f16x2") == 0)) return 2;
    if (m && (strcmp(m, "0") == 0 || strcmp(m, "f32") == 0)) return 0;
    if (m && (strcmp(m, "1") == 0 || strcmp(m, "bf16x3") == 0)) return 1;
    if (m && m[0]) fprintf(stderr, "libekaid_hip: unknown EKAID_GEMM_MODE '%s' (f32 | bf16x3 | f16x2): using the default\n", m);
    return EKAID_GEMM_MODE_DEFAULT;
}
DeviceState& device_state() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    DeviceState& s = g_dev[(d < 0 || d >= MAX_DEVICES) ? 0 : d];
    int expected = 0;
    if (s.ready.compare_exchange_strong(expected, 1)) {
        int unset = -1;
        s.gemm_mode.compare_exchange_strong(unset, default_gemm_mode());
    }
    return s;
}
int gemm_mode() {
    DeviceState& s = device_state();
    int m = s.gemm_mode.load();
    if (m < 0) { m = default_gemm_mode(); s.gemm_mode.store(m); }
    return m;
}
int set_gemm_mode(int mode) { device_state().gemm_mode.store(mode); return EKAID_OK; }
static int default_seq_mode() {
    const char* m = getenv("EKAID_SEQ_MODE");        // "stepwise" / "0": the launch-per-product chain; "persistent" / "1": one launch per sequence
    if (m && (m[0] == 's' || m[0] == '0')) return 0;
    if (m && (m[0] == 'p' || m[0] == '1')) return 1;
    return EKAID_SEQ_MODE_DEFAULT;
}
int seq_mode() {
    DeviceState& s = device_state();
    int m = s.seq_mode.load();
    if (m < 0) { m = default_seq_mode(); s.seq_mode.store(m); }
    return m;
}
const unsigned long long* dropout_epoch_ptr() { return device_state().epoch.load(std::memory_order_relaxed); }
double* device_scratch(size_t bytes) {
    // allocated once per device (first use, or ekaid_init), on the calling thread's current device; a second thread racing
    // here frees its own allocation and adopts the winner's
    DeviceState& s = device_state();
    double* p = s.scratch.load();
    if (p) return p;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    double* expected = nullptr;
    if (!s.scratch.compare_exchange_strong(expected, p)) { (void)hipFree(p); p = expected; }
    return p;
}
unsigned* seq_status_ptr() {
    DeviceState& s = device_state();
    unsigned* p = s.seq_status.load();
    if (p) return p;
    if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    unsigned* expected = nullptr;
    if (!s.seq_status.compare_exchange_strong(expected, p)) { (void)hipFree(p); p = expected; }
    return p;
}
// keyed by the kernel's own address: every template instantiation is its own entry (numbered slots handed out by hand collided
// once two families grew into each other's range)
bool func_attr_once(const void* kernel) {
    DeviceState& s = device_state();
    std::lock_guard<std::mutex> lock(s.attr_mu);
    for (const void* p : s.attr_done)
        if (p == kernel) return false;
    s.attr_done.push_back(kernel);
    return true;
}

static std::atomic<long long> g_counts[EKAID_COUNT_N];
void count_launch(int which) {
    if (which >= 0 && which < EKAID_COUNT_N) g_counts[which].fetch_add(1, std::memory_order_relaxed);
}
long long launch_count(int which) { return (which >= 0 && which < EKAID_COUNT_N) ? g_counts[which].load(std::memory_order_relaxed) : -1; }

__global__ void epoch_advance_kernel(unsigned long long* e, unsigned long long by) { e[0] += by; }
}  // namespace ekaid

extern "C" int ekaid_init(int device_id) {
    if (device_id < 0 || device_id >= ekaid::MAX_DEVICES) { ekaid::set_error("ekaid_init: bad device id %d", device_id); return EKAID_ERR_ARG; }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(device_id) != hipSuccess) {
        (void)hipGetLastError();
        ekaid::set_error("ekaid_init: cannot select device %d", device_id);
        return EKAID_ERR_ARG;
    }
    (void)ekaid::device_state();
    const bool ok = ekaid::device_scratch(ekaid::EKAID_SCRATCH_BYTES) != nullptr && ekaid::seq_status_ptr() != nullptr;
    (void)hipSetDevice(cur);
    if (!ok) { ekaid::set_error("ekaid_init: cannot allocate the reduction scratch on device %d", device_id); return EKAID_ERR_LAUNCH; }
    return EKAID_OK;
}
extern "C" int ekaid_set_dropout_epoch_ptr(const unsigned long long* epoch_dev) {
    ekaid::device_state().epoch.store(epoch_dev);
    return EKAID_OK;
}
extern "C" int ekaid_dropout_epoch_advance(unsigned long long* epoch_dev, unsigned long long by, void* stream) {
    if (!epoch_dev) { ekaid::set_error("ekaid_dropout_epoch_advance: null pointer"); return EKAID_ERR_ARG; }
    hipLaunchKernelGGL(ekaid::epoch_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, epoch_dev, by);
    return ekaid::check_launch("dropout_epoch_advance");
}

extern "C" int ekaid_set_seq_mode(int mode) {
    if (mode != 0 && mode != 1) { ekaid::set_error("ekaid_set_seq_mode: 0 (step-wise) or 1 (persistent)"); return EKAID_ERR_ARG; }
    ekaid::device_state().seq_mode.store(mode);
    return EKAID_OK;
}
extern "C" int ekaid_get_seq_mode(void) { return ekaid::seq_mode(); }

extern "C" int ekaid_seq_status_read(unsigned* host_value, void* stream) {
    if (!host_value) { ekaid::set_error("ekaid_seq_status_read: null pointer"); return EKAID_ERR_ARG; }
    unsigned* p = ekaid::seq_status_ptr();
    if (!p) { ekaid::set_error("ekaid_seq_status_read: no status word"); return EKAID_ERR_LAUNCH; }
    // through a pinned word: a copy into pageable memory takes the runtime's staged path (tens of microseconds per read, and
    // greedy_decode reads once per answer batch)
    static unsigned* pinned = nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!pinned && hipHostMalloc((void**)&pinned, sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pinned = nullptr; }
    unsigned* dst = pinned ? pinned : host_value;
    if (hipMemcpyAsync(dst, p, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        ekaid::set_error("ekaid_seq_status_read: copy failed");
        return EKAID_ERR_LAUNCH;
    }
    *host_value = *dst;
    return EKAID_OK;
}
extern "C" int ekaid_seq_status_clear(void* stream) {
    unsigned* p = ekaid::seq_status_ptr();
    if (!p || hipMemsetAsync(p, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); ekaid::set_error("ekaid_seq_status_clear failed"); return EKAID_ERR_LAUNCH; }
    return EKAID_OK;
}

namespace ekaid { long long launch_count(int which); }
extern "C" long long ekaid_launch_count(int which) { return ekaid::launch_count(which); }

extern "C" int ekaid_version(void) { return EKAID_ABI_VERSION; }
extern "C" const char* ekaid_last_error(void) { return ekaid::g_err; }
