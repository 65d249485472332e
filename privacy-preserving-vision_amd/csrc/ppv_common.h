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

// Once-per-DEVICE guard for function attributes (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property: a process-wide
// flag would leave a second GPU of the same process without it).  Racing threads may both set the attribute: idempotent.
struct PpvDevOnce {
    unsigned done_mask = 0;
    bool need() const {
        int d = 0;
        (void)hipGetDevice(&d);
        return !(__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & (1u << (d & 31)));
    }
    void done() {
        int d = 0;
        (void)hipGetDevice(&d);
        __atomic_fetch_or(&done_mask, 1u << (d & 31), __ATOMIC_RELEASE);
    }
};
// a failed attribute call is a status code of the entry point, not a silently failing launch later
#define PPV_ATTR(call)                         \
    do {                                       \
        if (hipError_t e_ = (call)) return -(int)e_; \
    } while (0)

extern "C" {
// exp(-2 pi i t / N) tables (float2 / double2) resident on the current device; created once per (device, N)
// under a mutex with a blocking copy -- call ppv_init() before stream capture.
const void* ppv_twiddles_f32(int N);
const void* ppv_twiddles_f64(int N);
}
