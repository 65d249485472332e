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

#ifdef __cplusplus
}
#endif
#endif
