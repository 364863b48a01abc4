// cu_hog - occupy N whole CUs for a fixed time (tools/cu_contention.py): one 1024-thread workgroup with 160 KB of LDS per CU, so
// nothing that needs LDS fits beside it; every wave sleeps in a loop bounded by the constant 100 MHz clock, then the grid drains.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void cu_hog_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x & 1023] = (char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (lds[(threadIdx.x * 7) & 1023] == 123 && ticks == 0) sink[0] = 1;
}

// light = 1: a workgroup the size of a communication kernel's (512 threads, 16 KB of LDS): it shares its CU with whatever still fits
extern "C" int cu_hog_light(int n_wgs, double milliseconds, int* sink, void* stream) {
    if (n_wgs <= 0) return 0;
    hipLaunchKernelGGL(cu_hog_kernel, dim3(n_wgs), dim3(512), 16 * 1024, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5), sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int cu_hog(int n_cus, double milliseconds, int* sink, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        attr = true;
    }
    if (n_cus <= 0) return 0;
    hipLaunchKernelGGL(cu_hog_kernel, dim3(n_cus), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5), sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
