// Shared helpers for libpamnet_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pamnet_hip.h"

#define PAMNET_WAVE 64

#define PAMNET_LAUNCH_CHECK()                                  \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return (int)e__;                \
    } while (0)

// Clears the thread's sticky last-error (torch's own probing calls can leave one behind) and returns the stream.
static inline hipStream_t as_stream(pamnet_stream_t s) {
    (void)hipGetLastError();
    return reinterpret_cast<hipStream_t>(s);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
