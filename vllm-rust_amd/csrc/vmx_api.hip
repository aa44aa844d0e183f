// Runtime entry points of libvmx: device selection, ABI version, thread-local errors.
#include "vmx_internal.h"

#include <mutex>

int vmx_attention_scratch_init(int device);   // attention_kernels.hip
int vmx_iq_packed_init(int device);          // iq_packed.hip

namespace {
thread_local char g_err[512] = "";
}

void vmx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Compute units of the current device (256 on MI355X); queried once per process, no device work.
int vmx_num_cus(void) {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
        return cus;
    }();
    return n;
}

extern "C" const char *vmx_last_error(void) { return g_err; }

extern "C" int vmx_abi_version(void) { return 1; }

extern "C" int vmx_init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        vmx_set_error("vmx_init: no HIP device visible (%s)", hipGetErrorString(e));
        return VMX_ELAUNCH;
    }
    VMX_REQUIRE(device >= 0 && device < count, "vmx_init: device %d out of range (count %d)", device, count);
    VMX_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VMX_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vmx_set_error("vmx_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return VMX_EUNSUPPORTED;
    }
    if (int rc = vmx_attention_scratch_init(device)) return rc;
    return vmx_iq_packed_init(device);
}
