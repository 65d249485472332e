// Adam step of a whole parameter list in ONE launch (gfx950).
//
// Reference: the harness' optimisers, Image_Caption/train.py:92-101 (`torch.optim.Adam(params=..., lr=...)` for the encoder and the
// decoder, stepped at train.py:318-321).  torch's fused Adam walks the ~320 tensors of the ResNet-101 trunk with multi_tensor_apply
// launches of 45-240 workgroups: 629 us per step for 1.19 GB (1.9 TB/s), and the step is byte-bound, so that time is not hidden by
// running it beside the next camera forward (PPV_OPT_OVERLAP=0 costs 0.1 ms).  Here every 4096-element chunk of every tensor is one
// workgroup of one launch (10.4 k workgroups for the trunk), the four operand streams of a chunk are all requested before the
// first store, and the arithmetic is torch's (`_fused_adam`, ADAM mode, no amsgrad / maximize), in f32:
//     g' = g + weight_decay * p;  m = m + (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;
//     p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps)
// with the bias corrections 1 - beta^step evaluated in double on the host.
#include "ppv_common.h"
#include "ppv_hip.h"

namespace ppv {

struct AdamDesc { float* p; const float* g; float* m; float* v; long numel; int blk0; int vec; };   // 48 bytes; vec: all four 16-byte aligned
constexpr int ADAM_NT = 256, ADAM_IT = 4, ADAM_CHUNK = ADAM_NT * ADAM_IT * 4;                    // 4096 elements per workgroup

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float step_size, float b1c, float b2, float b2c, float eps,
                                          float wd, float bc2_sqrt) {
    if (wd != 0.f) g = __builtin_fmaf(wd, p, g);
    m = m + b1c * (g - m);
    v = b2 * v + b2c * g * g;
    const float denom = __builtin_sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
    return p;
}

__global__ __launch_bounds__(ADAM_NT) void adam_multi_kernel(const AdamDesc* __restrict__ desc, int ndesc, float step_size, float b1c,
                                                             float beta2, float b2c, float eps, float wd, float bc2_sqrt) {
    int lo = 0, hi = ndesc - 1;                                 // the tensor that owns this block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamDesc d = desc[lo];
    const long base = (long)(blockIdx.x - d.blk0) * ADAM_CHUNK;
    if (d.vec && base + ADAM_CHUNK <= d.numel) {                // whole chunk, 16-byte operands: 16 loads in flight per thread
        float4 p[ADAM_IT], g[ADAM_IT], m[ADAM_IT], v[ADAM_IT];
#pragma unroll
        for (int it = 0; it < ADAM_IT; ++it) {
            const long i = base + (long)(it * ADAM_NT + threadIdx.x) * 4;
            p[it] = *reinterpret_cast<const float4*>(d.p + i);
            g[it] = *reinterpret_cast<const float4*>(d.g + i);
            m[it] = *reinterpret_cast<const float4*>(d.m + i);
            v[it] = *reinterpret_cast<const float4*>(d.v + i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < ADAM_IT; ++it) {
            const long i = base + (long)(it * ADAM_NT + threadIdx.x) * 4;
            adam_one(p[it].x, g[it].x, m[it].x, v[it].x, step_size, b1c, beta2, b2c, eps, wd, bc2_sqrt);
            adam_one(p[it].y, g[it].y, m[it].y, v[it].y, step_size, b1c, beta2, b2c, eps, wd, bc2_sqrt);
            adam_one(p[it].z, g[it].z, m[it].z, v[it].z, step_size, b1c, beta2, b2c, eps, wd, bc2_sqrt);
            adam_one(p[it].w, g[it].w, m[it].w, v[it].w, step_size, b1c, beta2, b2c, eps, wd, bc2_sqrt);
            *reinterpret_cast<float4*>(d.p + i) = p[it];
            *reinterpret_cast<float4*>(d.m + i) = m[it];
            *reinterpret_cast<float4*>(d.v + i) = v[it];
        }
        return;
    }
    for (long i = base + threadIdx.x; i < base + ADAM_CHUNK && i < d.numel; i += ADAM_NT) {        // ragged tail / unaligned tensor
        float p = d.p[i], m = d.m[i], v = d.v[i];
        adam_one(p, d.g[i], m, v, step_size, b1c, beta2, b2c, eps, wd, bc2_sqrt);
        d.p[i] = p; d.m[i] = m; d.v[i] = v;
    }
}

}  // namespace ppv

using namespace ppv;

extern "C" {

// desc: device array of ndesc records {float* p; const float* g; float* m; float* v; long numel; int blk0; int vec} (48 bytes each;
// blk0 = prefix sum of ceil(numel / 4096), vec = 1 when the four pointers are 16-byte aligned); total_blocks = the sum.  One Adam step
// (torch.optim.Adam, amsgrad = maximize = False) of every listed f32 tensor; bias_correction{1,2} = 1 - beta{1,2}^step.
// (hyper-parameters as doubles: 1 - beta2 formed from a float beta2 = 0.999f is off by 1.3e-5 relative)
int ppv_adam_multi(const void* desc, int ndesc, int total_blocks, double lr, double beta1, double beta2, double eps, double weight_decay,
                   double bias_correction1, double bias_correction2, hipStream_t stream) {
    if (!desc) return PPV_ERR_NULL;
    if (ndesc < 1 || total_blocks < 1 || !(bias_correction1 > 0.0) || !(bias_correction2 > 0.0)) return PPV_ERR_BAD_SIZE;
    const float step_size = (float)(lr / bias_correction1), bc2_sqrt = (float)sqrt(bias_correction2);
    adam_multi_kernel<<<total_blocks, ADAM_NT, 0, stream>>>((const AdamDesc*)desc, ndesc, step_size, (float)(1.0 - beta1), (float)beta2,
                                                            (float)(1.0 - beta2), (float)eps, (float)weight_decay, bc2_sqrt);
    return ppv_last_error();
}

}  // extern "C"
