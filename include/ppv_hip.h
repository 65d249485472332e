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
int ppv_fftconv_partials_per_image(int C, int N, int mode);
/* normalisation: Lens.py:312 (one group = whole batch) / Optics.py:128 (one group per image) */
int ppv_group_max(const float* partial, float* out, int groups, int per_group, ppv_stream_t stream);
int ppv_div_by_group(float* x, const float* m, long per_group, int groups, ppv_stream_t stream);

/* backward of the IC sensor image: Lens.py:290,312 + Utils.py:251-297 */
size_t ppv_fftconv_bwd_workspace_bytes(int B, int C, int N);
int ppv_sensor_dot_count(const float* g, const float* sensor, double* dotcnt, long n, ppv_stream_t stream);
int ppv_fftconv_ic_bwd(const float* img, const float* g_sensor, const float* sensor, const void* signs,
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
int ppv_ic_psf_state_offsets(int RR, int P, int K, size_t* off_h, size_t* off_F0, size_t* off_U, size_t* off_I32,
                             size_t* off_raw);

/* ---- Zernike basis (poppy.zernike.zernike_basis, IC Utils.py:75-77 / FD Utils.py:60-63) ---------------------
 * terms: K device records {int n, m, off, cnt; double norm}; coefs: device doubles of the radial polynomials. */
int ppv_zernike_basis(const void* terms, const double* coefs, float* out, int K, int npix, double scale,
                      double outside, ppv_stream_t stream);
int ppv_zernike_max_order(void);

#ifdef __cplusplus
}
#endif
#endif
