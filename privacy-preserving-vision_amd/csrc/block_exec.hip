// Whole-bottleneck launchers: one C-ABI call enqueues every kernel of a torchvision Bottleneck (Image_Caption/models.py:17-21 under
// train-mode BatchNorm, train.py:245) in the order ppv_amd/encoder.py enqueues them one by one.  Nothing new runs on the device --
// the functions below only call the entry points of conv_gemm.hip / trunk_ops.hip / conv_wgrad_stem.hip -- what changes is the host:
// the Python step crosses ctypes ~1000 times (15 ms per 22-ms step on a fast host, 23 ms on a slow one: the step then runs at the
// speed of the interpreter); per bottleneck that is 6 crossings forward and ~12 backward, here one each.
#include <hip/hip_runtime.h>
#include "ppv_common.h"
#include "ppv_hip.h"

extern "C" {

// forward of a bottleneck WITHOUT projection shortcut, statistics folded by the apply kernels (ppv_bn_act_fold_rows)
int ppv_bottleneck_fwd(const PpvBottleneckFwd* a, hipStream_t stream) {
    if (!a) return PPV_ERR_NULL;
    const int B = a->B, H = a->H, W = a->W, Cin = a->Cin, P = a->planes, st = a->stride;
    if (st != 1 && st != 2) return PPV_ERR_BAD_SIZE;
    const int H2 = H / st, W2 = W / st, C3 = 4 * P;
    if (Cin != C3) return PPV_ERR_BAD_SIZE;                 // identity shortcut: same width and (stride 1) same map
    if (st != 1) return PPV_ERR_BAD_SIZE;
    const long M1 = (long)B * H * W, M2 = (long)B * H2 * W2;
    int e;
    // conv1 1x1 -> bn1 + ReLU
    if ((e = ppv_conv_gemm(a->xin, a->w1, a->x1, a->stats1, nullptr, nullptr, a->zero_page, B, H, W, Cin, H, W, P, 1, 1, 1, 0, 1, 0, a->T1, stream))) return e;
    if ((e = ppv_bn_act_fold_rows(a->x1, a->stats1, a->T1, (double)M1, a->g1, a->b1, a->rm1, a->rv1, a->mom1, a->eps1, a->coef1, nullptr, a->y1,
                                  nullptr, M1 * P, P, 0, 1, stream))) return e;
    // conv2 3x3 (stride on the 3x3: torchvision v1.5) -> bn2 + ReLU
    if ((e = ppv_conv_gemm(a->y1, a->w2, a->x2, a->stats2, nullptr, nullptr, a->zero_page, B, H, W, P, H2, W2, P, 3, 3, st, -1, 1, 0, a->T2, stream))) return e;
    if ((e = ppv_bn_act_fold_rows(a->x2, a->stats2, a->T2, (double)M2, a->g2, a->b2, a->rm2, a->rv2, a->mom2, a->eps2, a->coef2, nullptr, a->y2,
                                  nullptr, M2 * P, P, 0, 1, stream))) return e;
    // conv3 1x1 -> bn3 + identity + ReLU (+ the (y > 0) bit mask the backward pass reads)
    if ((e = ppv_conv_gemm(a->y2, a->w3, a->x3, a->stats3, nullptr, nullptr, a->zero_page, B, H2, W2, P, H2, W2, C3, 1, 1, 1, 0, 1, 0, a->T3, stream))) return e;
    return ppv_bn_act_fold_rows(a->x3, a->stats3, a->T3, (double)M2, a->g3, a->b3, a->rm3, a->rv3, a->mom3, a->eps3, a->coef3, a->xin, a->yout,
                                a->bits, M2 * C3, C3, 1, 1, stream);
}

}  // extern "C"
