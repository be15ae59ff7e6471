// Development probe (tools/cu_contention_probe.py): K workgroups that do nothing but hold their CU for a given time -- a stand-in
// for the RCCL all-reduce kernels that share the device with the backward at N > 1 (their workgroups take CUs the step's
// one-workgroup-per-CU kernels count on).  Never part of libpamnet_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void squat_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();                 // 100 MHz
    long long t = t0;
    int v = 0;
    while (t - t0 < ticks) {
        for (int i = 0; i < 64; ++i) v += __builtin_amdgcn_readfirstlane(v + i);
        t = wall_clock64();
    }
    if (v == 0x7fffffff) *sink = v;
}
extern "C" int squat(int wgs, double micros, int* sink, void* stream) {
    hipLaunchKernelGGL(squat_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (long long)(micros * 100.0), sink);
    return (int)hipGetLastError();
}
