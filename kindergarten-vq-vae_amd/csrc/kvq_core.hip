// kvq_core.hip -- error state, version and device info of libkvq.so
#include <string.h>

#include "kvq_common.h"

namespace kvq {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return KVQ_OK;
}

}  // namespace kvq

extern "C" {

int kvq_version(void) { return KVQ_VERSION; }

const char* kvq_last_error(void) { return kvq::g_err; }

int kvq_device_info(int* cu_count, char* name, size_t name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return kvq::fail(KVQ_E_NODEVICE, "hipGetDevice failed: no usable HIP device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return kvq::fail(KVQ_E_NODEVICE, "hipGetDeviceProperties failed");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return KVQ_OK;
}

}  // extern "C"
