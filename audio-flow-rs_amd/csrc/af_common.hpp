// af_common.hpp -- shared host-side helpers for libaudioflow_dsp.so
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/audioflow_dsp.h"

namespace af {

// thread-local message behind af_last_error()
std::string& last_error_slot();
af_status fail(af_status code, const char* fmt, ...);

// Makes sure a gfx950 device is usable; never aborts.
af_status ensure_device(int device /* -1 = current */);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Current device ordinal (0 when the runtime cannot say) and its CU count.
inline int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev;
}
int device_cu_count(int device);

// Diagnostic switches behind af_debug_set() (include/audioflow_dsp.h); 0 = product behaviour.
enum DebugKnob {
    kKnobFeaturesKernel = 0,   // 1: the general fused kernel also at hop 256
    kKnobFeaturesWgPerCu,      // 1, 2: workgroups per CU of the fused kernels
    kKnobBandedOff,            // 1: resampler / FIR on the thread-per-output kernels
    kKnobChainUnfused,         // 1: the chain runs resampler and FIR as two launches
    kKnobCount
};
int debug_knob(DebugKnob k);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: `done` is the
// caller's static per-kernel flag array, indexed by device ordinal.
constexpr int kMaxDevices = 64;
// (atomic flags: handles on different host threads launch the same kernel concurrently; setting the attribute
// twice is harmless, a torn flag is not something to reason about)
typedef std::atomic<bool> LdsAttrFlags[kMaxDevices];
template <class K>
inline hipError_t allow_big_lds(K kern, int bytes, LdsAttrFlags& done)
{
    const int dev = current_device();
    if (dev < kMaxDevices && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev < kMaxDevices) done[dev].store(true, std::memory_order_release);
    return e;
}

#define AF_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess)                                                     \
            return ::af::fail(_e == hipErrorOutOfMemory ? AF_ERR_OUT_OF_MEMORY    \
                                                        : AF_ERR_HIP,             \
                              "%s failed: %s (%s:%d)", #expr,                     \
                              hipGetErrorString(_e), __FILE__, __LINE__);         \
    } while (0)

#define AF_REQUIRE(cond, ...)                                                     \
    do {                                                                          \
        if (!(cond)) return ::af::fail(AF_ERR_INVALID_ARG, __VA_ARGS__);          \
    } while (0)

#define AF_TRY(expr)                                                              \
    do {                                                                          \
        af_status _s = (expr);                                                    \
        if (_s != AF_OK) return _s;                                               \
    } while (0)

// RAII device buffer (host-side bookkeeping only)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only reserve
    af_status reserve(size_t n)
    {
        if (n <= bytes) return AF_OK;
        release();
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(AF_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", n,
                        hipGetErrorString(e));
        }
        bytes = n;
        return AF_OK;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// features_api.hip: the feature kernel on the units of clips [clip_begin, clip_end) of the handle's batch (rows land at
// their places in mel_dev), with wg_per_cu workgroups per CU (1 | 2) -- the chain's pipelined form
// whether the handle's batches run as ONE launch of a fused kernel (no dense filterbank contraction, no to_mono pass)
bool features_one_launch(const af_features* p);
af_status features_run_clips(af_features* p, const void* pcm_dev, float* mel_dev, hipStream_t stream, size_t clip_begin,
                             size_t clip_end, int wg_per_cu);

}  // namespace af
