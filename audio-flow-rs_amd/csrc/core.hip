// core.hip -- error reporting and device bring-up for libaudioflow_dsp.so
#include <atomic>
#include <cstring>
#include <mutex>

#include "af_common.hpp"

namespace af {

std::string& last_error_slot()
{
    static thread_local std::string slot;
    return slot;
}

af_status fail(af_status code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

af_status ensure_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(AF_ERR_NO_DEVICE,
                    "no HIP device available (%s); libaudioflow_dsp has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device >= 0) {
        if (device >= n) return fail(AF_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
        AF_HIP(hipSetDevice(device));
    }
    int cur = 0;
    AF_HIP(hipGetDevice(&cur));
    hipDeviceProp_t prop;
    AF_HIP(hipGetDeviceProperties(&prop, cur));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(AF_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", cur,
                    prop.gcnArchName);
    return AF_OK;
}

namespace {
std::atomic<int> g_knobs[kKnobCount];
const char* const kKnobNames[kKnobCount] = {"features.kernel", "features.wg_per_cu", "banded.off",
                                            "chain.unfused"};
}  // namespace

int debug_knob(DebugKnob k) { return g_knobs[k].load(std::memory_order_relaxed); }

int device_cu_count(int device)
{
    // queried once per device ordinal; 256 (MI355X) only if the runtime cannot say
    static std::mutex mu;
    static int cus[kMaxDevices] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < kMaxDevices && cus[device] > 0) return cus[device];
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        n = prop.multiProcessorCount;
    if (device >= 0 && device < kMaxDevices) cus[device] = n;
    return n;
}

}  // namespace af

extern "C" {

const char* af_last_error(void) { return af::last_error_slot().c_str(); }

const char* af_version(void) { return "audioflow-dsp-mi355x 0.1.0 (gfx950)"; }

af_status af_debug_set(const char* key, int value)
{
    if (!key) return af::fail(AF_ERR_INVALID_ARG, "af_debug_set: null key");
    for (int i = 0; i < af::kKnobCount; ++i)
        if (std::strcmp(key, af::kKnobNames[i]) == 0) {
            af::g_knobs[i].store(value, std::memory_order_relaxed);
            return AF_OK;
        }
    return af::fail(AF_ERR_INVALID_ARG, "af_debug_set: unknown switch '%s'", key);
}

int af_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
