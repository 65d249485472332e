// Shared host-side helpers of libppv_hip.so (status codes, twiddle tables).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define PPV_OK 0
#define PPV_ERR_NULL (-1001)      // required pointer is null
#define PPV_ERR_BAD_SIZE (-1002)  // unsupported shape
#define PPV_ERR_INIT (-1003)      // table / plan creation failed
#define PPV_ERR_WORKSPACE (-1004) // workspace too small

// 0 or -(hipError_t) of the last launch on this thread
static inline int ppv_last_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PPV_OK : -(int)e;
}

extern "C" {
// exp(-2 pi i t / N) tables (float2 / double2) resident on the current device; created once per (device, N)
// under a mutex with a blocking copy -- call ppv_init() before stream capture.
const void* ppv_twiddles_f32(int N);
const void* ppv_twiddles_f64(int N);
}
