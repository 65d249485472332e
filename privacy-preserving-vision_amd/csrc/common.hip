// Immutable per-device constant tables for libppv_hip.so.
#include "ppv_common.h"
#include <cmath>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {
std::mutex g_mu;
std::map<std::pair<int, int>, void*> g_tw32, g_tw64;

template <typename T2, typename T>
const void* get_table(std::map<std::pair<int, int>, void*>& cache, int N) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = cache.find({dev, N});
    if (it != cache.end()) return it->second;
    std::vector<T2> h(N);
    for (int t = 0; t < N; ++t) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)N;
        h[t].x = (T)cosl(a);
        h[t].y = (T)sinl(a);
    }
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(T2) * N) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(T2) * N, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    cache[{dev, N}] = d;
    return d;
}
}  // namespace

extern "C" {

const void* ppv_twiddles_f32(int N) { return get_table<float2, float>(g_tw32, N); }
const void* ppv_twiddles_f64(int N) { return get_table<double2, double>(g_tw64, N); }

// Create every table the library can need on the current device (blocking). Returns 0 / <0.
int ppv_init(void) {
    const int n32[] = {256, 512, 1024};
    for (int n : n32)
        if (!ppv_twiddles_f32(n)) return PPV_ERR_INIT;
    return PPV_OK;
}

int ppv_abi_version(void) { return 21; }   // 21: ppv_bilinear_resize_fwd / _bwd; 20: ppv_bn_f32_*, ppv_maxpool_f32_*, ppv_adaptive_pool_f32_*, ppv_conv3x3_bnin, ppv_ic_psf_set_fields_f32; 19: ppv_mse_fwd, ppv_mse_bwd; 18: ppv_adam_multi; 17: ppv_gemm_bf16x3_tn, ppv_decc_enc_grad, ppv_decc_mean, ppv_fftconv_ic_fwd_p / _bwd_p (any even patch size <= 512); 16: ppv_gemm_f32_tn; 15: ppv_gemm_f32_ws, ppv_gemm_f32_ws_plan; 14: ppv_conv_bn_relu_coop (experiment); 13: ppv_ic_psf_state_init, ppv_im2col_split, ppv_pad_split; 12: ppv_trunk_fwd, ppv_trunk_bwd, ppv_stream_fork; 11: ppv_conv_wgrad_ex, ppv_wgrad_reduce_multi; 10: ppv_bottleneck_bwd; 9: ppv_bottleneck_fwd; 8: ppv_bn_act_fold_rows; 7: ppv_ic_psf_symmetric; 6: ppv_ic_psf_mark_support; 2: mask_bits / pos_bits, decoder, SSIM, alt-corr, split-bf16 entries; 3: ppv_conv_gemm_red; 4: ppv_stem_dgrad, ppv_bn_bwd_sums2; 5: uint8 image entries (ppv_fftconv_fwd_u8, ppv_fftconv_ic_bwd_u8)

}  // extern "C"
