/* libppv_hip.so -- C ABI of the MI355X-native Camera + ResNet-101 hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its boundary is the
 * nn.Module surface (Camera.Lens.OpticsZernike, Camera.Optics.Camera, models.Encoder), which the
 * Python package keeps; these flat functions are what those modules call per fused stage.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch tensors); the library never
 *    allocates or frees user-visible memory; scratch comes from a caller workspace
 *    (ppv_*_workspace_bytes()).  Only immutable twiddle tables are created internally
 *    (mutex-guarded, once per device; call ppv_init() before stream capture).
 *  - every function takes the HIP stream to launch on and returns int: 0 ok, <0 error
 *    (-hipError_t, or PPV_ERR_* below).  Nothing throws, exits or synchronises the device.
 *  - re-entrant; callable from the autograd engine's worker thread.
 */
#ifndef PPV_HIP_H
#define PPV_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ppv_stream_t;   /* == hipStream_t */

#define PPV_ERR_NULL (-1001)
#define PPV_ERR_BAD_SIZE (-1002)
#define PPV_ERR_INIT (-1003)
#define PPV_ERR_WORKSPACE (-1004)

int ppv_abi_version(void);
int ppv_init(void);

/* ---- FFT image (x) PSF convolution ------------------------------------------------------------
 * Replaces Image_Caption/Camera/Utils.py:251-297 img_psf_conv (+ psf2otf :127-158) and
 * Face-DeId/Camera/Utils.py:7-12 conv2D.  N = FFT length (512 or 256).
 * mode 0 (IC): img [B,C,P,P], P = N/2, zero-padded linear conv, |.|, crop, nearest P-1 -> P map.
 * mode 1 (FD): img [B,C,N,N], circular conv with the PSF rolled by -N/2 (Optics.py:126). */
size_t ppv_fftconv_workspace_bytes(int B, int C, int N);
size_t ppv_otf_elems(int C, int N);                       /* float2 elements of OTF^T [C][N/2+1][N] */
int ppv_otf_build(const void* psf, int psf_is_f64, long sc, long sy, long sx, int C, int P, int N,
                  void* otfT, void* workspace, ppv_stream_t stream);
int ppv_fftconv_fwd(const float* img, const void* otfT, float* out, void* signs, float* partial_max,
                    void* workspace, int B, int C, int N, int mode, int conj_otf, ppv_stream_t stream);
/* the same on the data set's uint8 pixels, decoded as x / 255 inside the row transform (Image_Caption/datasets.py:46
 * `imgs / 255.`; HDF5 uint8 [N,3,256,256], utils.py:94-150): the f32 copy of the batch never exists */
int ppv_fftconv_fwd_u8(const unsigned char* img, const void* otfT, float* out, void* signs, float* partial_max,
                       void* workspace, int B, int C, int N, int mode, int conj_otf, ppv_stream_t stream);
int ppv_fftconv_partials_per_image(int C, int N, int mode);
/* normalisation: Lens.py:312 (one group = whole batch) / Optics.py:128 (one group per image) */
int ppv_group_max(const float* partial, float* out, int groups, int per_group, ppv_stream_t stream);
int ppv_div_by_group(float* x, const float* m, long per_group, int groups, ppv_stream_t stream);

/* backward of the IC sensor image: Lens.py:290,312 + Utils.py:251-297 */
/* The IC sensor convolution for ANY even patch size P <= 512 (Lens.py:21-22: the constructor's default is 368) on an N-point transform,
 * N = 256 / 512 / 1024 >= 2 P: the P x P supports make every N >= 2 P - 1 compute the reference's 2 P-point result (Utils.py:251-297).
 * signs [B * C][P][N / 128] 64-bit words, partial_max ppv_fftconv_ic_partials(B, C, P) floats, u8: uint8 pixels decoded as x / 255. */
int ppv_fftconv_ic_partials(int B, int C, int P);
size_t ppv_fftconv_ic_workspace_bytes(int B, int C, int P, int N);
int ppv_fftconv_ic_fwd_p(const void* img, int u8, const void* otfT, float* out, void* signs, float* partial_max, void* workspace,
                         int B, int C, int P, int N, ppv_stream_t stream);
size_t ppv_fftconv_ic_bwd_workspace_bytes_p(int B, int C, int P, int N);
int ppv_fftconv_ic_bwd_p(const void* img, int u8, const float* g_sensor, const float* sensor, const void* signs, const float* maxv,
                         const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64, long sc, long sy, long sx,
                         float* g_img, void* workspace, const void* sx_saved, int B, int C, int P, int N, ppv_stream_t stream);
/* sx_saved (may be null): the first planes * P * (N / 2) float2 of the workspace ppv_fftconv_ic_fwd_p ran on (the row transform of the
 * image), if the caller kept that buffer untouched since: backward then skips recomputing it */
size_t ppv_fftconv_bwd_workspace_bytes(int B, int C, int N);
int ppv_sensor_dot_count(const float* g, const float* sensor, double* dotcnt, long n, ppv_stream_t stream);
int ppv_fftconv_ic_bwd(const float* img, const float* g_sensor, const float* sensor, const void* signs,
                       const float* maxv, const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64,
                       long sc, long sy, long sx, float* g_img, void* workspace, int B, int C, int N,
                       ppv_stream_t stream);
/* uint8 pixels (datasets.py:46): no gradient w.r.t. the image, g_img must be NULL */
int ppv_fftconv_ic_bwd_u8(const unsigned char* img, const float* g_sensor, const float* sensor, const void* signs,
                          const float* maxv, const double* dotcnt, const void* otfT, void* g_psf, int g_psf_is_f64,
                          long sc, long sy, long sx, float* g_img, void* workspace, int B, int C, int N,
                          ppv_stream_t stream);

/* ---- IC PSF generation: Image_Caption/Camera/Lens.py:158-274 (+ Utils.py:80-109,192-248,328-413) ---------
 * Z [K][RR][RR] f32 basis, coeffs [K] f32, noise [RR*RR] f32 U[0,1) (Utils.py:403), sph [RR][RR][3] c64
 * (Lens.py:191-210, cached), Ht [3][M][M] c64 Fresnel transfer function stored kx-major (Utils.py:339-373, cached),
 * kdn[3] HOST doubles 2*pi/lambda*(n-1), m1/m2 [P][P][3] f64 masks (Lens.py:111-127) or NULL.
 * Outputs: psf_n [P][P][3] f32 (Lens.py:239), psf_m f64 (Lens.py:274), loss_acc = sum of squares (Lens.py:271). */
size_t ppv_ic_psf_state_bytes(int RR, int P, int K);
int ppv_ic_psf_fwd(const float* Z, const float* coeffs, const float* noise, const void* sph, const void* Ht,
                   const double* kdn, float tol, const double* m1, const double* m2, float* psf_n, double* psf_m,
                   double* loss_acc, void* state, int RR, int P, int K, int up, float up_scale, ppv_stream_t stream);
int ppv_ic_psf_bwd(const float* Z, const void* Ht, const double* kdn, const double* m1, const double* m2,
                   const float* psf_n, const double* g_psf_m, const float* g_psf_n, const double* g_loss,
                   const double* loss, float* g_coeffs, void* state, int RR, int P, int K, int up, float up_scale,
                   ppv_stream_t stream);
/* Required once for a fresh state buffer, before its first use: clears the marks below (memory recycled from an earlier state could
 * otherwise carry a valid-looking mark of another basis). */
int ppv_ic_psf_state_init(void* state, int RR, int P, int K, ppv_stream_t stream);
/* Optional, once per (state, Z) -- the mark is bound to the address of Z and to K, and ignored (full passes) for any other basis
 * buffer; re-mark after writing into Z in place: marks the support of the basis inside `state` so that the two calls above skip the pixel groups
 * where every plane of Z is zero (outside the aperture disk of poppy.zernike_basis(outside=0), Utils.py:75-77; exact). */
int ppv_ic_psf_mark_support(const float* Z, void* state, int RR, int P, int K, ppv_stream_t stream);
/* 1 when ppv_ic_psf_mark_support found the basis bitwise mirror-symmetric (every Noll term is even or odd under x -> -x and y -> -y on
 * poppy's centred grid, Utils.py:75-77), in which case ppv_ic_psf_fwd / _bwd read one quadrant of it; 0 otherwise.  Synchronises
 * `stream`: a test / diagnostic entry point, not called in the step. */
int ppv_ic_psf_symmetric(const void* state, int RR, int P, int K, ppv_stream_t stream);
int ppv_ic_psf_state_offsets(int RR, int P, int K, size_t* off_h, size_t* off_F0, size_t* off_U, size_t* off_I32,
                             size_t* off_raw);
/* 1: the Fresnel transforms and the saved fields F0 / U are c64 (default since round 5: the reference feeds its transform c64-valued
 * fields, Utils.py:80-85, so c64 adds ~5e-7 against the 1e-3 tolerance), 0: c128 (PPV_PSF_F32=0) */
int ppv_ic_psf_fields_f32(void);
/* 1 / 0: c64 / c128 Fresnel fields from now on (-1: the environment variable PPV_PSF_F32 decides again); returns the previous setting.
 * The state layout depends on it: choose BEFORE ppv_ic_psf_state_bytes / ppv_ic_psf_state_init of a state and keep it while that state
 * lives (a module-level choice; the reference's fields are complex128 by type promotion, Image_Caption/Camera/Utils.py:328-378). */
int ppv_ic_psf_set_fields_f32(int on);

/* ---- Zernike basis (poppy.zernike.zernike_basis, IC Utils.py:75-77 / FD Utils.py:60-63) ---------------------
 * terms: K device records {int n, m, off, cnt; double norm}; coefs: device doubles of the radial polynomials. */
int ppv_zernike_basis(const void* terms, const double* coefs, float* out, int K, int npix, double scale,
                      double outside, ppv_stream_t stream);
int ppv_zernike_max_order(void);

/* ---- ResNet-101 trunk (Image_Caption/models.py:13-41 Encoder -> torchvision ResNet-101) -----
 * NHWC bf16 implicit-GEMM convolution on MFMA; serves forward and data-gradient (see csrc/conv_gemm.hip).
 *   forward: a = stride, off = -pad, div = 1, Wt = [Cout][R][S][Cin];
 *   dgrad:   a = 1, off = -(k-1-pad), div = stride, Wt = [Cin][R][S][Cout] taps flipped.
 * stat_part [stat_rows][2][N]: PRE-ZEROED partial (sum, sum of squares) of the bf16-rounded outputs (train-mode BN;
 * row tiles fold into row tile % stat_rows with f32 atomics; ppv_conv_stat_tiles(M) gives stat_rows) or NULL; addend [M][N] bf16 is added before rounding (residual-gradient accumulation) or NULL;
 * mask_bits [M*N/8] bytes or NULL (bit k of byte i <-> element 8i+k, written by ppv_bn_act's pos_bits): output lanes whose
 * bit is clear are stored as 0 (the ReLU backward of the tensor the data gradient flows into, torchvision Bottleneck
 * `out = relu(out + identity)`, folded into the store at 1/16 of the mask tensor's bytes). */
int ppv_conv_gemm(const void* X, const void* Wt, void* out, float* stat_part, const void* addend, const void* mask_bits,
                  const void* zero_page, int B, int Hs, int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int a,
                  int off, int div, int out_f32, int stat_rows, ppv_stream_t stream);
/* conv2 of an identity bottleneck with bn1 + ReLU applied INSIDE the convolution (round 6: the LDS-resident halo tile of conv_halo.hip is
 * activated once per 64-channel chunk; the element-wise launch between conv1 and conv2 disappears).  Train-mode BatchNorm of
 * Image_Caption/train.py:245 on the torchvision Bottleneck behind Image_Caption/models.py:17-21.  x_raw [B,H,W,C] bf16 = conv1's raw
 * output, sums [T][2][C] its partial sums (the stat_part of that ppv_conv_gemm call), count = B*H*W.  Leaves what ppv_bn_act_fold_rows
 * left: coef [4][C] (scale, shift, mean, invstd), running statistics updated (run_mean / run_var may be null), y_act [B,H,W,C] bf16 =
 * relu(bn(x_raw)) (the operand of conv2's weight gradient; null: not written); out [B,H,W,N] bf16 = conv3x3(y_act), stride 1, pad 1,
 * wt [N][3][3][C], with its own statistics in stat_part [stat_rows][2][N] (PRE-ZEROED).  ppv_conv3x3_bnin_supported: 1 where this form
 * runs (16- / 32-wide maps, N % 128 == 0, C % 64 == 0, C <= 512, launches that fill the chip): callers keep the two-launch form
 * elsewhere (ppv_bottleneck_fwd uses it under PPV_BNIN=1). */
int ppv_conv3x3_bnin_supported(int B, int H, int W, int C, int N);
int ppv_conv3x3_bnin(const void* x_raw, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                     float* run_var, float momentum, float eps, float* coef, void* y_act, const void* wt, void* out, float* stat_part,
                     int stat_rows, const void* zero_page, int B, int H, int W, int C, int N, ppv_stream_t stream);

/* data-gradient launch that also takes the sums the following BatchNorm backward needs (torch's batch_norm_backward reduce):
 * red_part [red_rows][2][N] f32, PRE-ZEROED, receives sum g and sum g * red_x per column, g = the tensor as stored (addend
 * added, rounded to bf16, masked), red_x [M][N] bf16 = the raw convolution output that BatchNorm normalised.  bf16 output,
 * N % 128 == 0; ppv_bn_bwd(..., part_prezeroed = 2) with red_rows = 8 then skips its own reduce pass over g and x.
 * red_coef (NULL, or that BatchNorm's coef rows [scale | shift] when a ReLU without residual follows it; not with addend):
 * lanes with x * scale + shift <= 0 are stored as 0 and left out of the sums -> ppv_bn_bwd runs with relu = 0. */
int ppv_conv_gemm_red(const void* X, const void* Wt, void* out, float* red_part, const void* red_x, const float* red_coef,
                      const void* addend, const void* mask_bits, const void* zero_page, int B, int Hs, int Ws, int Cs, int Ho, int Wo, int N,
                      int R, int S, int a, int off, int div, int red_rows, ppv_stream_t stream);
/* RAFT SepConvGRU (Face-DeId/RAFT/core/update.py:33-77): stride-1 convolution with a rectangular kernel and separate row / column
 * paddings (1 x 5 with (0, 2), 5 x 1 with (2, 0)); the gate arithmetic between the convolutions, NHWC f32 */
int ppv_conv_gemm_rect(const void* X, const void* Wt, void* out, const void* zero_page, int B, int H, int W, int Cs, int N, int R,
                       int S, int pad_h, int pad_w, int out_f32, ppv_stream_t stream);
int ppv_gru_zr(const float* zr, int ldzr, const float* bias, const float* h, float* z, float* rh, long rows, int Ch,
               ppv_stream_t stream);
int ppv_gru_out(const float* q, int ldq, const float* bias, const float* z, const float* h, float* hn, long rows, int Ch,
                ppv_stream_t stream);
/* Dense layers of the caption decoder (Image_Caption/models.py:199-214) in exact f32 on v_mfma_f32_16x16x4_f32:
 * out[m][n] = sum_k x[m][k] W[n][k] + bias[n]; x [M][K] (row stride ldx), W [N][K] (row stride ldw), bias [N] or NULL, out row stride
 * ldo.  K % 16 == 0.  ksplit > 1 ADDS partial sums with f32 atomics into a PRE-ZEROED out (ppv_gemm_f32_ksplit gives the count). */
int ppv_gemm_f32_ksplit(int M, int N, int K);
int ppv_gemm_f32(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                 int ksplit, ppv_stream_t stream);
/* the same product with the K slices written to workspace [ksplit][M][N] f32 and summed in index order by a second small launch: no
 * atomics, no pre-zeroed output, bit-reproducible for any split.  ppv_gemm_f32_ws_plan returns the split to pass (1: no workspace
 * needed) and the workspace bytes. */
int ppv_gemm_f32_ws_plan(int M, int N, int K, size_t* bytes);
/* out[m][n] = sum_k a[k][m] b[k][n] (both operands K-major, any K): the batched weight gradients g^T h of the decoder's dense layers
 * (models.py:199-214 autograd) without transposed copies; split and workspace from ppv_gemm_f32_tn_plan */
int ppv_gemm_f32_tn_plan(int M, int N, int K, size_t* bytes);
int ppv_gemm_f32_tn(const float* a, long lda, const float* b, long ldb, float* out, long ldo, int M, int N, int K, int ksplit,
                    void* workspace, ppv_stream_t stream);
int ppv_gemm_f32_ws(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                    int ksplit, void* workspace, ppv_stream_t stream);
/* ppv_gemm_f32_tn's product a^T b (same contract) on the BF16 matrix pipe: the f32 operands are split in the kernel into bf16 hi + lo and
 * a product is hi*hi + hi*lo + lo*hi with f32 accumulation (~1e-5 of sum |a b|; 5.3x the matrix rate of the exact-f32 form).  The
 * default of the decoder's five batched weight gradients (replaces the autograd GEMMs of Image_Caption/models.py:199-214). */
/* the row-major sibling: out = x W^T + bias (ppv_gemm_f32's contract) as three bf16 products of in-kernel splits -- the decoder's two large
 * non-recurrent products (vocabulary layer over all time steps, models.py:211, and its transposed data gradient); K % 4 == 0 */
int ppv_gemm_bf16x3_nt_plan(int M, int N, int K, size_t* bytes);
int ppv_gemm_bf16x3_nt(const float* x, long ldx, const float* W, long ldw, const float* bias, float* out, long ldo, int M, int N, int K,
                       int ksplit, void* workspace, ppv_stream_t stream);
int ppv_gemm_bf16x3_tn_plan(int M, int N, int K, size_t* bytes);
int ppv_gemm_bf16x3_tn(const float* a, long lda, const float* b, long ldb, float* out, long ldo, int M, int N, int K, int ksplit,
                       void* workspace, ppv_stream_t stream);
/* InstanceNorm2d / AdaIN (+ LeakyReLU) of the StarGAN-v2 blocks (Face-DeId/core/model.py:12-124), NHWC f32:
 * y = lrelu((x - mean_bc) * invstd_bc * scale + shift), statistics per (sample, channel) over HW, eps as given; scale / shift are
 * [C] (per_sample = 0: nn.InstanceNorm2d(affine=True)) or [B][C] (per_sample = 1: AdaIN's (1 + gamma), beta).  stats / sums:
 * [B][C] float2 = (mean, invstd) / (d shift, d scale) per (sample, channel). */
int ppv_instnorm_fwd(const float* x, const float* scale, const float* shift, float* y, void* stats, int B, int HW, int C,
                     int per_sample, float slope, float eps, ppv_stream_t stream);
int ppv_instnorm_bwd(const float* x, const float* g, const void* stats, const float* scale, const float* shift, float* dx,
                     void* sums, int B, int HW, int C, int per_sample, float slope, ppv_stream_t stream);
/* EXPERIMENT (round 4; measured in profiles/r04*_coop_ab.json, not on the product path): 1x1 / unit-stride convolution + train-mode
 * BatchNorm2d + ReLU (torchvision Bottleneck conv1 -> bn1 -> relu, Image_Caption/models.py:17-21 under train.py:245) in ONE launch: the
 * 256 x 128 tiles, one workgroup per CU, leave their statistics, cross a grid barrier and normalise the tile they still hold in LDS.
 * x_raw and y [B,H,W,N] bf16 are both written; stats [stat_rows][2][N] f32 and counter[2] PRE-ZEROED (counter[1] != 0 afterwards: a
 * workgroup gave up waiting -- results invalid); coef [4][N] as ppv_bn_finalize.  PPV_ERR_BAD_SIZE unless the whole grid can be resident. */
int ppv_conv_bn_relu_coop(const void* X, const void* Wt, void* x_raw, void* y, float* stats, unsigned* counter, const float* gamma,
                          const float* beta, float* run_mean, float* run_var, float momentum, float eps, float* coef,
                          const void* zero_page, int B, int H, int W, int Cs, int N, int stat_rows, ppv_stream_t stream);
int ppv_conv_stat_tiles(long M);
int ppv_conv_set_variant(int v);   /* tuning hook: 0 auto, 1 two-stage, 2 128x128x4-stage, 3 256x128x3-stage */
int ppv_weight_layout_multi(const void* desc, int ndesc, int total_blocks, ppv_stream_t stream);
int ppv_weight_layout(const float* w, void* out, int Cout, int Cin, int R, int S, int mode, ppv_stream_t stream);

/* ---- optimiser (csrc/optim.hip) ------------------------------------------------------------------
 * One Adam step (torch.optim.Adam semantics, amsgrad = maximize = False; the encoder / decoder optimisers of
 * /root/reference/Image_Caption/train.py:92-101, stepped at :318-321) of a whole list of f32 tensors in one launch.
 * desc: DEVICE array of ndesc 48-byte records {float* p; const float* g; float* m; float* v; long numel; int blk0; int vec}:
 * blk0 = prefix sum of ceil(numel / 4096) over the records, vec = 1 when the four pointers are 16-byte aligned;
 * total_blocks = that sum.  bias_correction{1,2} = 1 - beta{1,2}^step (the caller counts steps). */
int ppv_adam_multi(const void* desc, int ndesc, int total_blocks, double lr, double beta1, double beta2, double eps, double weight_decay,
                   double bias_correction1, double bias_correction2, ppv_stream_t stream);

/* weight gradient (layer2..4 trainable, models.py:43-54): torch layout [N][Cs][R][S] f32 out; per-slice slabs in scratch */
size_t ppv_conv_wgrad_scratch_bytes(long M, int N, int R, int S, int Cs);
int ppv_conv_wgrad(const void* G, const void* X, float* dW_out, void* scratch, const void* zero_page, int B, int Hs,
                   int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int stride, int pad, ppv_stream_t stream);
/* The same with the slab reduce optionally left to the caller (several weight gradients of one bottleneck reduced by ONE launch):
 * deferred != NULL and a kernel form with accumulator-order slabs -> no reduce launch, *deferred describes it (blocks > 0) and
 * `scratch` must stay intact until ppv_wgrad_reduce_multi has run; blocks == 0: the call finished the gradient itself. */
typedef struct PpvWgradReduce {
    const void* slabs; float* out;
    int N, C, R, S, nslab, TN, mode, blocks;
} PpvWgradReduce;
int ppv_conv_wgrad_ex(const void* G, const void* X, float* dW_out, void* scratch, const void* zero_page, int B, int Hs,
                   int Ws, int Cs, int Ho, int Wo, int N, int R, int S, int stride, int pad, ppv_stream_t stream, PpvWgradReduce* deferred);
int ppv_wgrad_reduce_multi(const PpvWgradReduce* probs, int n, ppv_stream_t stream);

/* Two 1x1 / unit-stride weight gradients of different shapes in ONE launch + one reduce launch (round 6: conv1 of the bottleneck whose
 * backward just finished and conv3 of the next one; each problem takes half the chip with m-slices twice as long -- half the slabs, half
 * the per-workgroup fixed cost per unit of work).  G[k] [B,H[k],W[k],N[k]] bf16, X[k] [B,H[k],W[k],Cs[k]] bf16 -> dW[k] [N[k]][Cs[k]]
 * f32; N[k] % 256 == 0, Cs[k] % 128 == 0.  scratch: scratch_bytes bytes, no zeroing (PPV_ERR_WORKSPACE when too small: the two problems'
 * ppv_conv_wgrad_scratch_bytes added are always enough).  ppv_conv_wgrad_pair_supported: 1 where this form runs. */
int ppv_conv_wgrad_pair_supported(int B, int H0, int W0, int Cs0, int N0, int H1, int W1, int Cs1, int N1);
int ppv_conv_wgrad_pair(const void* G0, const void* X0, float* dW0, int H0, int W0, int Cs0, int N0, const void* G1, const void* X1, float* dW1,
                        int H1, int W1, int Cs1, int N1, void* scratch, size_t scratch_bytes, const void* zero_page, int B, ppv_stream_t stream);

/* P <= 24 weight gradients of ONE 1x1 / unit-stride shape in one launch, each reduced over all its rows by one workgroup per
 * tile (no split-M slabs, no scratch, no reduce launch): G[p] [B,H,W,N] bf16, X[p] [B,H,W,Cs] bf16 -> out[p] [N][Cs] f32.
 * Host arrays of device pointers.  Replaces P cuDNN weight-gradient calls of the bottleneck 1x1 convolutions behind
 * Image_Caption/models.py:17-21 (autograd of torchvision Bottleneck.conv1 / conv3).  N % 128 == 0, Cs % 128 == 0. */
int ppv_conv_wgrad_group(const void* const* G, const void* const* X, float* const* out, int P, const void* zero_page, int B, int H,
                         int W, int Cs, int N, hipStream_t stream);
int ppv_wgrad_set_variant(int v);  /* tuning hook, see csrc/conv_wgrad_stem.hip */
/* stem 7x7/2 conv (resnet.0), f32 NCHW sensor image in, NHWC bf16 out; data gradient via ppv_conv_gemm (N = 16) */
int ppv_stem_weight_layout(const float* w, void* out, int mode, ppv_stream_t stream);
int ppv_stem_conv(const float* img, const void* wst, void* out, float* stat_part, int stat_rows, int B, int H,
                  int W, ppv_stream_t stream);
int ppv_stem_dgrad_scatter(const float* t, float* g, int B, int Ho, int Wo, ppv_stream_t stream);
/* the whole stem data gradient in one launch (Wo % 128 == 0): g_raw [B,Ho,Wo,64] bf16 -> g_img [B,3,2Ho,2Wo] f32 NCHW;
 * zero_page: >= 128 zero bytes (source of the out-of-image taps) */
int ppv_stem_dgrad(const void* g_raw, const void* wsd, float* g_img, const void* zero_page, int B, int Ho, int Wo,
                   ppv_stream_t stream);
/* train-mode BatchNorm2d (+ residual, + ReLU), forward and backward */
int ppv_bn_finalize(const float* part, int T, double count, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float momentum, float eps, float* coef, int C, ppv_stream_t stream);
int ppv_bn_act(const void* x, const float* coef1, const void* r, const float* coef2, void* y, void* pos_bits, long n, int C,
               int res_mode, int relu, long res_mod, ppv_stream_t stream);

/* torch BatchNorm2d(training) + ReLU (+ identity add) of torchvision's Bottleneck (Image_Caption/models.py:17-21, train.py:245) with the
 * statistics in ONE row sums [2][C] (ppv_conv_gemm with stat_rows = 1): every thread derives scale / shift of its eight channels
 * itself (no ppv_bn_finalize launch); coef [4][C] (scale, shift, mean, invstd) and the running statistics (momentum, unbiased
 * variance; run_mean / run_var may be NULL) are written by the threads of the first row.  res_mode 0: none, 1: identity r. */
/* ---- whole-bottleneck launchers (host-side only: they call the entry points above/below in the order the Python step does) ----------
 * One crossing of the FFI per torchvision Bottleneck (Image_Caption/models.py:17-21, train-mode BatchNorm: train.py:245) instead of
 * one per kernel.  Forward of a block WITHOUT projection shortcut: conv1 1x1 -> bn1+ReLU -> conv2 3x3 -> bn2+ReLU -> conv3 1x1 ->
 * bn3 + identity + ReLU, every BatchNorm as ppv_bn_act_fold_rows (statistics in T partial rows, PRE-ZEROED by the caller; running
 * statistics updated in place; coef [4][C] written for the backward pass).  All tensors NHWC bf16; bits = (yout > 0) mask, n/8 bytes. */
typedef struct PpvBottleneckFwd {
    const void *xin, *w1, *w2, *w3;                       /* block input [B,H,W,4 planes]; forward weight layouts (ppv_weight_layout mode 0) */
    void *x1, *y1, *x2, *y2, *x3, *yout, *bits;           /* raw conv outputs, activations, block output, its sign mask */
    float *stats1, *stats2, *stats3;                      /* [T][2][C] f32, zeroed */
    float *coef1, *coef2, *coef3;                         /* [4][C] f32: scale, shift, mean, invstd */
    const float *g1, *b1; float *rm1, *rv1;               /* BatchNorm weight, bias, running_mean, running_var */
    const float *g2, *b2; float *rm2, *rv2;
    const float *g3, *b3; float *rm3, *rv3;
    const void* zero_page;                                /* >= 256 zero bytes (padding source of the convolutions) */
    float mom1, eps1, mom2, eps2, mom3, eps3;
    int B, H, W, Cin, planes, stride, T1, T2, T3;         /* Cin == 4 * planes, stride == 1 (identity shortcut) */
} PpvBottleneckFwd;
int ppv_bottleneck_fwd(const PpvBottleneckFwd* a, ppv_stream_t stream);
/* Backward of the same block (the default schedule of ppv_amd/encoder.py): bn3' -> dgrad3 -> bn2' -> dgrad2 -> bn1' -> dgrad1 on `main`,
 * each weight gradient on `side` (null: on `main`) behind an event recorded on `main` right after the BatchNorm backward that produces its
 * operand.  g: gradient w.r.t. the block output, already masked by that output's ReLU.  part3: bn3's sums, part3_ready != 0 when the
 * data-gradient launch that produced g took them (ppv_conv_gemm_red), else PRE-ZEROED scratch; part2 / part1 PRE-ZEROED (>= 64 C floats each);
 * red2 / red1: the conv3 / conv2 data gradients also take bn2's / bn1's sums (shapes ppv_conv_gemm_red supports).  x3_prev / part3_prev
 * (may be null): raw conv3 output and PRE-ZEROED sums buffer of the block the returned gradient flows into.  dg / db / dw null = that
 * parameter is frozen (models.py:43-54).  kc: 3 C floats of scratch per BatchNorm.  wscratch: ppv_conv_wgrad_scratch_bytes. */
typedef struct PpvBottleneckBwd {
    const void *g, *xin, *x1, *y1, *x2, *y2, *x3, *xin_bits;
    const float *c1, *c2, *c3;
    const void *wd1, *wd2, *wd3;                          /* data-gradient weight layouts (ppv_weight_layout mode 1) */
    float *part3, *part2, *part1, *kc3, *kc2, *kc1;
    void *gx3, *gy2, *gx2, *gy1, *gx1, *gin;              /* bf16 gradients: raw conv3 / act2 / raw conv2 / act1 / raw conv1 / block input */
    float *dg3, *db3, *dg2, *db2, *dg1, *db1, *dw3, *dw2, *dw1;
    void* wscratch;
    const void* x3_prev; float* part3_prev;
    const void* zero_page;
    int B, H, W, planes, part3_ready, red2, red1;
    long wstride;                                         /* > 0: wscratch holds three regions wstride bytes apart, the block's slab reduces run as one launch */
} PpvBottleneckBwd;
int ppv_bottleneck_bwd(const PpvBottleneckBwd* a, ppv_stream_t main_stream, ppv_stream_t side_stream);

/* ---- whole-trunk executor (csrc/trunk_plan.hip): the forward of models.py:31-41's ResNet-101 trunk (train-mode BatchNorm, train.py:245)
 * from ONE call and its backward from ONE call, over one caller-provided arena -- the launches and their order are those of the
 * per-kernel entry points above (stem, projection bottlenecks, identity bottlenecks through ppv_bottleneck_fwd / _bwd, pools, weight
 * gradients forked to `side`, final join).  The arena's layout is a pure function of PpvTrunkDesc: [zero zone: BatchNorm partial sums
 * of the step, cleared with one memset per direction | what forward keeps for backward | per-block gradient buffers | scratch].
 * A forward invalidates what the previous forward on the same arena kept. */
#define PPV_TRUNK_MAX_BLOCKS 64
typedef struct PpvTrunkBlock {
    int planes, stride, proj;      /* torchvision Bottleneck(inplanes, planes, stride); proj != 0: downsample = conv1x1(stride) + BatchNorm */
    int train_w;                   /* bit 0 / 1 / 2 / 3: conv1 / conv2 / conv3 / shortcut weight is trainable (sizes the slab scratch) */
} PpvTrunkBlock;
typedef struct PpvTrunkDesc {
    int B, H, W, nblocks;          /* images [B,3,H,W]; H, W multiples of 32 */
    int fold_rows;                 /* partial rows of the statistics the apply kernels fold themselves (ppv_bn_act_fold_rows) */
    int wgrad_reduce3;             /* 1: the three slab reduces of an identity bottleneck as one launch */
    int _r0, _r1;
    PpvTrunkBlock blk[PPV_TRUNK_MAX_BLOCKS];
} PpvTrunkDesc;
typedef struct PpvTrunkConv {      /* one convolution + its BatchNorm: [0] = stem, [1 + 4 b + k] = conv1 / conv2 / conv3 / shortcut of block b */
    const void *wt, *wd;           /* bf16 forward / data-gradient layouts (ppv_weight_layout modes 0 / 1; stem: ppv_stem_weight_layout) */
    const float *gamma, *beta; float *rm, *rv;     /* BatchNorm weight, bias, running_mean, running_var (the last two may be null) */
    float *dw, *dgamma, *dbeta;    /* gradient destinations (torch layouts, f32); null = frozen (models.py:43-54) */
} PpvTrunkConv;
typedef struct PpvTrunkHyper { float mom, eps; } PpvTrunkHyper;     /* same indexing as PpvTrunkConv */
size_t ppv_trunk_arena_bytes(const PpvTrunkDesc* d);                 /* 0: unsupported geometry */
int ppv_trunk_block_offsets(const PpvTrunkDesc* d, int blk, size_t* out20);
int ppv_trunk_fwd(const PpvTrunkDesc* d, const PpvTrunkConv* cv, const PpvTrunkHyper* hy, const float* images, void* arena,
                  void* cells_out, const void* zero_page, ppv_stream_t stream);
/* blocks [blk_lo, blk_hi) in reverse order; blk_hi == nblocks: g_top = gradient of the output (g_kind 0: bf16 [B,h,w,C] already masked by
 * the last ReLU; 1 / 2: f32 / bf16 [B,E,E,C] gradient of AdaptiveAvgPool2d(E)'s output, models.py:27,39); blk_lo == 0: the stem follows
 * (g_img [B,3,H,W] f32 or null) and main_stream waits for side_stream.  cells: the cells_out of the forward call. */
int ppv_trunk_bwd(const PpvTrunkDesc* d, const PpvTrunkConv* cv, void* arena, const void* cells, const void* g_top, int g_kind, int E,
                  float* g_img, const void* zero_page, int blk_lo, int blk_hi, ppv_stream_t main_stream, ppv_stream_t side_stream);
/* event record on `from` + wait on `to` (a guarded ring of timing-less events per device) */
int ppv_stream_fork(ppv_stream_t from, ppv_stream_t to);
/* a stream restricted to CUs [first_cu, first_cu + n_cus) of the 256-bit CU mask (bits are dealt round-robin over the XCDs) */
int ppv_stream_create_masked(ppv_stream_t* out, int first_cu, int n_cus);
int ppv_stream_destroy(ppv_stream_t s);

int ppv_bn_act_fold(const void* x, const float* sums, double count, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                    int res_mode, int relu, ppv_stream_t stream);
/* as ppv_bn_act_fold with the statistics in T partial rows sums [T][2][C] (ppv_conv_gemm, stat_rows = T): fewer adders per address in the
 * convolution's epilogue than one row, still no ppv_bn_finalize launch.  C / 8 a power of two <= 256 when T > 1. */
int ppv_bn_act_fold_rows(const void* x, const float* sums, int T, double count, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float momentum, float eps, float* coef, const void* r, void* y, void* pos_bits, long n, int C,
                    int res_mode, int relu, ppv_stream_t stream);

/* Train-mode BatchNorm in ONE launch: partial statistics [T][2][C] (as ppv_conv_gemm leaves them) -> per-channel coefficients (written
 * to coef [4][C] = scale, shift, mean, invstd for the backward pass; running statistics updated with `momentum`, unbiased variance)
 * -> y = act(x * scale + shift (+ residual)).  res_mode 0: none, 1: identity r, 2: r normalised by its own BatchNorm (the *2
 * arguments: the projection shortcut's).  pos_bits as ppv_bn_act.  Replaces ppv_bn_finalize + ppv_bn_act for
 * torch BatchNorm2d(training) + ReLU (+ add) of torchvision's Bottleneck (Image_Caption/models.py:17-21, train.py:245).
 * C % 64 == 0 for C <= 256, else C % 256 == 0. */
int ppv_bn_act_train(const void* x, const float* part, int T, double count, const float* gamma, const float* beta, float* run_mean,
                     float* run_var, float momentum, float eps, float* coef, const void* r, const float* part2, int T2,
                     const float* gamma2, const float* beta2, float* run_mean2, float* run_var2, float momentum2, float eps2,
                     float* coef2, void* y, void* pos_bits, long rows, int C, int res_mode, int relu, hipStream_t stream);
int ppv_bn_bwd_blocks(long rows, int C);
int ppv_bn_bwd(const void* gy, const void* y, const void* x, const float* coef, double count, void* gx, void* gpre,
               float* dgamma, float* dbeta, float* part, float* kc, long rows, int C, int relu, int part_prezeroed,
               ppv_stream_t stream);
/* ppv_bn_bwd (relu = 0) that also takes the backward sums of a second BatchNorm fed by the same gradient (the projection
 * shortcut of a down-sampling bottleneck, raw conv output x2) into part2 [8][2][C], PRE-ZEROED; that BatchNorm's ppv_bn_bwd then
 * runs with part_prezeroed = 2 */
int ppv_bn_bwd_sums2(const void* gy, const void* x, const float* coef, double count, void* gx, float* dgamma, float* dbeta,
                     float* part, float* kc, long rows, int C, int part_prezeroed, const void* x2, float* part2,
                     ppv_stream_t stream);
/* stem BN + ReLU + MaxPool 3x3/2 (resnet.1-3) and AdaptiveAvgPool2d(36) (models.py:27,39-40) */
int ppv_bn_relu_maxpool(const void* x, const float* coef, void* y, void* arg, int B, int H, int W, int C,
                        ppv_stream_t stream);
int ppv_maxpool_relu_bwd(const void* gy, const void* y, const void* arg, void* gpre, int B, int H, int W, int C,
                         ppv_stream_t stream);
/* backward of resnet.1-3 from the pooled gradient straight to the gradient of the raw stem convolution output (torch autograd of
 * MaxPool2d(3,2,1) o ReLU o BatchNorm2d(train), models.py:17-21): = ppv_maxpool_relu_bwd + ppv_bn_bwd without the pre-pool tensor.
 * part: >= 16 * C floats of scratch. */
int ppv_maxpool_bn_bwd(const void* gy, const void* y, const void* arg, const void* x, const float* coef, double count, void* gx,
                       float* dgamma, float* dbeta, float* part, int B, int H, int W, int C, ppv_stream_t stream);
int ppv_adaptive_pool_fwd(const void* x, void* y, int B, int H, int W, int C, int E, int out_f32, ppv_stream_t stream);
int ppv_adaptive_pool_bwd(const void* gy, void* gx, const void* mask_src, int B, int H, int W, int C, int E, int g_f32,
                          ppv_stream_t stream);

/* ---- FD camera PSF: Face-DeId/Camera/Optics.py:92-120 (+ losses :113,:124-125), complex64, N in {256, 512} ------
 * base = rad*(t*focus), chirp1, chirp3 [3][N][N] c64 and chirp2T [3][kx][ky] c64 are cached constants (Optics.py:94-107);
 * kf[3] HOST floats k*flmb (Optics.py:89-90).  acc[4] f64 = {sum (rho psf)^2, centering rows, centering cols, total}. */
int ppv_zernike_contract(const float* Z, const float* coeffs, float* h, int K, long npx, ppv_stream_t stream);
size_t ppv_fd_psf_workspace_bytes(int N);
int ppv_fd_psf_fwd(const float* h, const void* base, const void* chirp1, const void* chirp2T, const void* chirp3,
                   const float* rho, const float* kf, float lratio, float amp, float* psf, double* acc,
                   void* workspace, int N, ppv_stream_t stream);

/* ---- RAFT correlation block: Face-DeId/RAFT/core/corr.py:12-60 (fp32) -------------------------------------------------- */
int ppv_corr_volume(const float* f1, const float* f2, float* corr, int B, int C, int HW, ppv_stream_t stream);
int ppv_avgpool2(const float* in, float* out, long n, int H, int W, ppv_stream_t stream);
int ppv_corr_lookup_bwd(const float* gout, const float* coords, float* gcorr_l, int B, int H1, int W1, int Hl, int Wl,
                        int r, int level, int nlevels, ppv_stream_t stream);
int ppv_avgpool2_bwd_acc(const float* g_coarse, float* g_fine, long n, int H, int W, ppv_stream_t stream);
int ppv_corr_volume_bwd(const float* gcorr, const float* f1, const float* f2, float* g_f1, float* g_f2, int B, int C,
                        int HW, ppv_stream_t stream);
int ppv_corr_lookup(const float* corr_l, const float* coords, float* out, int B, int H1, int W1, int Hl, int Wl, int r,
                    int level, int nlevels, ppv_stream_t stream);
int ppv_corr_lookup_all(const float* const* corr_levels, const int* Hl, const int* Wl, int levels, const float* coords, float* out,
                        int B, int H1, int W1, int r, ppv_stream_t stream);   /* every level in one launch (host arrays) */

/* ---- FAN heat-map regressor forward, eval mode: Face-DeId/core/wing.py:178-260 (glue around ppv_conv_gemm) ------------- */
int ppv_stem_conv6(const float* img, const void* wst, void* out, int B, int H, int W, ppv_stream_t stream);
int ppv_fan_input(const float* x, const float* coords, float* out, int B, int Hin, int Win, int S, ppv_stream_t stream);
/* Bilinear resize of `planes` NCHW f32 planes [Hi][Wi] -> [Ho][Wo] with torch's F.interpolate(mode='bilinear') arithmetic (both
 * align_corners settings; scale = in / out resp. (in - 1) / (out - 1)) and its adjoint as a deterministic gather:
 * FAN.get_heatmap_train, Face-DeId/core/wing.py:264,270. */
int ppv_bilinear_resize_fwd(const float* x, float* y, long planes, int Hi, int Wi, int Ho, int Wo, int align_corners, ppv_stream_t stream);
int ppv_bilinear_resize_bwd(const float* gy, float* gx, long planes, int Hi, int Wi, int Ho, int Wo, int align_corners, ppv_stream_t stream);
int ppv_avgpool2_nhwc(const void* x, void* y, int B, int H, int W, int C, int f32, ppv_stream_t stream);   /* f32: 0 bf16 tensors, 1 f32 */
int ppv_upsample2_add(const void* up1, const void* low, void* out, int B, int H, int W, int C, int f32, ppv_stream_t stream);
int ppv_concat3_add(const void* o1, const void* o2, const void* o3, const void* res, void* out, long M, int n1, int n2,
                    int n3, int s1, int s2, int s3, int f32, ppv_stream_t stream);
/* bf16 split of an f32 matrix stacked along the rows: y [3*rows][Cp] = [hi; lo; hi] (mode 0) or [hi; hi; lo] (mode 1), so that
 * y0(g)^T y1(h) on ppv_conv_wgrad equals g^T h to ~2^-16: the decoder's batched weight gradients (models.py:199-214 autograd). */
/* Operands of the weight gradient of an f32 NHWC convolution whose channel counts do not fit ppv_conv_wgrad's 128-wide tiles (the
 * 3- / 64-channel layers of Face-DeId/core/model.py:12-53): one half of the bf16 split (part 0: hi = bf16(x), 1: lo = bf16(x - hi)) of
 * x [rows][C] zero-padded to Cp channels, or of its R x S patch rows [B*Ho*Wo][Kp] (column (r S + s) C + c; a 1x1 weight gradient over
 * Kp "channels" then gives the R x S x C one -- the K-padding of the trunk's stem).  Stacked along the batch as [hi; lo; hi] against
 * [hi; hi; lo] the bf16 MFMA weight gradient equals the f32 one to ~2^-16. */
int ppv_im2col_split(const float* x, void* out, int B, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad, int Kp,
                     int part, ppv_stream_t stream);
int ppv_pad_split(const float* x, void* out, long rows, int C, int Cp, int part, ppv_stream_t stream);
int ppv_split3_rows(const float* x, long ldx, void* y, long rows, int C, int Cp, int mode, ppv_stream_t stream);
int ppv_bn_act_split3(const float* x, const float* coef, void* y, long rows, int C, int Cp, int relu, int ldx, ppv_stream_t stream);
int ppv_fan_head(const void* raw, const float* bias, float* raw_out, float* sums, float* heat, int B, int S, int ldr,
                 int nch, int split, int nsum, int up, ppv_stream_t stream);

/* on-the-fly windowed correlation = Face-DeId/RAFT/core/corr.py:63-91 AlternateCorrBlock + the reference's CUDA extension
 * RAFT/alt_cuda_corr/correlation_kernel.cu:18-119 (forward) / :122-256 (backward), all pyramid levels per forward call (fmap2_levels / H2 / W2 are HOST arrays), N = 1
 * coordinate set per pixel; NHWC f32 maps, coords (x, y); no HW x HW volume is materialised */
int ppv_alt_corr_fwd(const float* fmap1, const float* const* fmap2_levels, const int* H2, const int* W2, int levels,
                     const float* coords, float* out, int B, int H1, int W1, int C, int r, float scale, ppv_stream_t stream);
int ppv_alt_corr_bwd(const float* fmap1, const float* fmap2, const float* coords, const float* gout, float* d_fmap1,
                     float* d_fmap2, int B, int H1, int W1, int H2, int W2, int C, int r, float scale, int level, int levels,
                     ppv_stream_t stream);

/* ---- SSIM loss, Image_Caption/pytorch_ssim/__init__.py:20-40 (camera_loss = 'SSIM', train.py:172-173): window 11, sigma 1.5,
 * zero padding; `win` = the 11 normalised 1-D taps (HOST array); forward accumulates per-image sums of the SSIM map */
int ppv_ssim_fwd(const float* img1, const float* img2, double* sums, const float* win, int B, int C, int H, int W, ppv_stream_t stream);
int ppv_ssim_bwd(const float* img1, const float* img2, const float* gscale, float* d_img2, const float* win, int B, int C, int H, int W,
                 ppv_stream_t stream);

/* ---- fp32 trunk, element-wise side (round 6: Encoder(precision="fp32")): train-mode BatchNorm2d (+ residual) (+ ReLU), the stem's
 * max-pool and the adaptive average pool on NHWC float32 activations, forward and backward; the torchvision ResNet-101 behind
 * Image_Caption/models.py:17-41 trained in fp32 (train.py:245).  Deterministic (per-row-block f64 partial sums added in block order).
 * ppv_bn_f32_fwd: x [rows][C] -> y = act(x * scale + shift + res) (res null: none), coef [4][C] OUT (scale, shift, mean, invstd: what
 * the backward call reads); train = 1: batch statistics (biased variance), running statistics updated (null: not tracked);
 * train = 0: the running statistics.  workspace: ppv_bn_f32_workspace_bytes(C) bytes, no zeroing.  C % 64 == 0.
 * ppv_bn_f32_bwd: g = dL/dy, y (only read behind a ReLU: the mask y > 0), x, coef of the forward call -> gx, gres (null: no residual
 * branch; else the masked gradient), dgamma / dbeta (null: not wanted), sums [2][C] OUT (d beta, d gamma).
 * ppv_maxpool_f32_*: 3 x 3, stride 2, padding 1; arg [B,Ho,Wo,C] u8 = window offset of the FIRST maximum in row-major order
 * (torch.nn.MaxPool2d's choice); backward is a gather (no atomics).  ppv_adaptive_pool_f32_*: AdaptiveAvgPool2d((E, E)), models.py:27.
 * ppv_split6_rows: x [rows][C] f32 -> y [rows][Cp] bf16 = [h | m | h | l | h | m | 0 ...] of the three-way bf16 split x = h + m + l: the
 * operand of an f32-level convolution on the bf16 MFMA kernels against the K-concatenated filter [H | H | M | H | L | M]
 * (ppv_amd.nn_ops.conv2d_f32(exact=True)).  C % 4 == 0, Cp % 4 == 0, Cp >= 6 C. */
size_t ppv_bn_f32_workspace_bytes(int C);
int ppv_split6_rows(const float* x, void* y, long rows, int C, int Cp, ppv_stream_t stream);
int ppv_bn_f32_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                   const float* res, float* y, float* coef, void* workspace, long rows, int C, int relu, int train, ppv_stream_t stream);
int ppv_bn_f32_bwd(const float* g, const float* y, const float* x, const float* coef, float* gx, float* gres, float* dgamma, float* dbeta,
                   float* sums, void* workspace, long rows, int C, int relu, int train, ppv_stream_t stream);
int ppv_maxpool_f32_fwd(const float* x, float* y, void* arg, int B, int H, int W, int C, ppv_stream_t stream);
int ppv_maxpool_f32_bwd(const float* gy, const void* arg, float* gx, int B, int H, int W, int C, ppv_stream_t stream);
int ppv_adaptive_pool_f32_fwd(const float* x, float* y, int B, int H, int W, int C, int E, ppv_stream_t stream);
int ppv_adaptive_pool_f32_bwd(const float* gy, float* gx, int B, int H, int W, int C, int E, ppv_stream_t stream);

/* ---- camera MSE loss of the harness, Image_Caption/train.py:170-171,284-288 (`loss_cam = 1 - nn.MSELoss()(imgs, sensor)`), fused.
 * ppv_mse_fwd: out[0] = mean((a - b)^2) in one pass, deterministic (per-workgroup f64 partials, summed in index order by a
 * one-workgroup launch); `workspace` = ppv_mse_workspace_bytes() bytes, 16-byte aligned (contents irrelevant).  ppv_mse_bwd: g_b = g_in + k (b - a), k = coef * gscalar[0] (gscalar: DEVICE scalar, the
 * gradient autograd hands to the loss; coef = 2 / n for mse, -2 / n for 1 - mse); g_in = the gradient that reaches b through its
 * other consumer (null: none), i.e. autograd's accumulation of the two gradients happens in the same pass; g_a (null: not wanted)
 * = -k (b - a).  a, b, g_* f32, n elements, 16-byte aligned. */
size_t ppv_mse_workspace_bytes(void);
int ppv_mse_fwd(const float* a, const float* b, long n, void* workspace, float* out, ppv_stream_t stream);
int ppv_mse_bwd(const float* g_in, const float* a, const float* b, const float* gscalar, float coef, float* g_b, float* g_a, long n,
                ppv_stream_t stream);

/* ---- soft-attention LSTM caption decoder, Image_Caption/models.py:57-218.  encoder_att is hoisted out
 * of the time loop (models.py:83 recomputes it every step) and runs through ppv_conv_gemm; these are the per-step kernels.
 * "sorted" = the batch order after the caption-length sort of models.py:181-183; order[b] = original image index. */
int ppv_dec_prepare(const float* enc, const long* order, void* encs, float* mean, int B, int P, int E, ppv_stream_t stream);
int ppv_dec_attend_fwd(const void* att1, const void* encs, const float* hproj, int ldh, const float* wfull, float* ebuf,
                       float* alpha_out, float* awe_save, float* xh, int ldx, int x_off, int bt, int P, int A, int E,
                       ppv_stream_t stream);
int ppv_dec_attend_bwd(const void* att1, const void* encs, const float* hproj, int ldh, const float* wfull, const float* alpha,
                       const float* awe_save, const float* dxh, int ldx, int x_off, const float* dalpha_in, float* dhproj,
                       float* dawe_out, float* dalpha, float* datt1, float* dwfull, int bt, int P, int A, int E,
                       ppv_stream_t stream);
int ppv_lstm_cell_fwd(const float* z, const float* c_prev, float* gates, float* c_new, float* h_a, int ld_a, float* h_b, int ld_b,
                      int bt, int D, ppv_stream_t stream);
int ppv_lstm_cell_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh, const float* dc_in, float* dz,
                      float* dc_prev, int bt, int D, ppv_stream_t stream);
int ppv_dec_enc_grad(const float* part, const float* dmean, const float* alpha, const float* dawe, const long* order, float* out,
                     int B, int P, int E, int T, ppv_stream_t stream);
/* per-image mean of the pooled tensor taken on the cell map: mean[b][e] = sum_c gamma[c] * cells[b][c][e] (cells bf16, models.py:143-145) */
int ppv_decc_mean(const void* cells, const float* gamma, float* mean, int B, int C, int E, ppv_stream_t stream);
/* the compact (cell-map) form: out[order[b]][c][e] = part + gamma[c] * dmean[b][e] + sum_t beta[t][b][c] * dawe[t][b][e], bf16 or f32 out */
int ppv_decc_enc_grad(const float* part, const float* dmean, const float* beta, const float* dawe, const long* order, const float* gamma,
                      void* out, int out_bf16, int B, int C, int E, int T, ppv_stream_t stream);
/* compact attention: the same step on the C cells of the map that AdaptiveAvgPool2d(36) (models.py:27,39) up-samples to the P
 * pixels -- class tables (cells / weight / multiplicity per distinct pixel class, class of every pixel) are built by the host */
int ppv_decc_attend_fwd(const void* att1c, const void* feat, const float* hproj, int ldh, const float* wfull, const int* cls_cells,
                        const float* cls_w, const float* cls_mult, const int* pix_class, float* alpha_out, float* alq_out,
                        float* beta_out, float* awe_save, float* xh, int ldx, int x_off, int bt, int P, int Q, int C, int A, int E,
                        ppv_stream_t stream);
int ppv_decc_attend_bwd(const void* att1c, const void* feat, const float* hproj, int ldh, const float* wfull, const int* cls_cells,
                        const float* cls_w, const float* cls_mult, const int* pix_class, const int* cell_cls, const float* alq,
                        const float* awe_save, const float* dxh, int ldx, int x_off, const float* galpha, float* dhproj,
                        float* dawe_out, float* dfb, float* datt1c, float* dwfull, int bt, int P, int Q, int C, int A, int E,
                        ppv_stream_t stream);
int ppv_dec_combine(const float* acc, const float* dmean, const long* order, float* out, int B, int P, int E, ppv_stream_t stream);

/* backward of the FD camera: sensor image -> PSF (Optics.py:126-128), PSF + losses -> height map (Optics.py:92-120),
 * height map -> Zernike coefficients (Optics.py:79-83) */
size_t ppv_fftconv_fd_bwd_workspace_bytes(int B, int C, int N);
int ppv_fftconv_fd_bwd(const float* img, const float* g_sensor, const float* sensor, const float* maxv, float* g_psf,
                       void* workspace, int B, int C, int N, ppv_stream_t stream);
size_t ppv_fd_psf_bwd_workspace_bytes(int N);
int ppv_fd_psf_bwd(const float* g_psf, const double* g_lr, const double* g_cl, const float* psf, const void* base,
                   const void* chirp1, const void* chirp2T, const void* chirp3, const float* rho, const float* kf,
                   float lratio, float amp, const float* h, const double* acc, float* gh, void* workspace,
                   void* workspace2, int N, ppv_stream_t stream);
size_t ppv_zernike_grad_scratch_bytes(int K, long npx);
int ppv_zernike_grad(const float* Z, const float* gh, float* g_coeffs, void* part, int K, long npx, ppv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
