// vo_runtime.hip -- error reporting, version and device check of libvo_hip.so.
#include "vo_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void vo_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vo_last_error(void) { return g_err; }

extern "C" int vo_version(void) { return 100; }

__global__ void vo_probe_kernel(int* out) { out[0] = 0x600d; }

extern "C" int vo_device_check(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        (void)hipGetLastError();
        vo_set_error("vo_device_check: no HIP device %d (count %d)", device, n);
        return VO_ENODEVICE;
    }
    hipDeviceProp_t prop;
    VO_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vo_set_error("vo_device_check: device %d is %s, this library is built for gfx950 only", device,
                     prop.gcnArchName);
        return VO_ENODEVICE;
    }
    int prev = 0;
    VO_HIP_TRY(hipGetDevice(&prev));
    VO_HIP_TRY(hipSetDevice(device));
    int* d = nullptr;
    int h = 0;
    VO_HIP_TRY(hipMalloc(&d, sizeof(int)));
    hipLaunchKernelGGL(vo_probe_kernel, dim3(1), dim3(1), 0, 0, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    (void)hipSetDevice(prev);
    if (e != hipSuccess || h != 0x600d) {
        vo_set_error("vo_device_check: probe kernel failed on device %d: %s", device, hipGetErrorString(e));
        return VO_ENODEVICE;
    }
    return VO_OK;
}
