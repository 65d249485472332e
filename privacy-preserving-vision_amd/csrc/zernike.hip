// Noll-ordered Zernike basis on the GPU (init-time constant of both cameras).
// Replaces the poppy.zernike.zernike_basis call of reference Image_Caption/Camera/Utils.py:75-77 and
// Face-DeId/Camera/Utils.py:60-63 (67 s on the host for 350 x 896^2) with one kernel:
//   Z_j(x,y) = norm_j * R_n^|m|(rho) * {cos|sin}(|m| theta) inside rho <= 1, `outside` elsewhere, times `scale`.
// fp64 evaluation, one rounding to f32 (the reference stores the basis as float32, Lens.py:69-77).
#include <hip/hip_runtime.h>
#include "ppv_common.h"

namespace ppv {

constexpr int ZMAXN = 40;    // highest radial order supported (K up to 861 terms)

// term table (host-built): for Noll index j: n, m (signed), norm, first coefficient offset; coefficients are for
// rho^(n - 2k), k = 0..(n-|m|)/2
struct ZTerm { int n, m, off, cnt; double norm; };

__global__ __launch_bounds__(256) void zernike_basis_kernel(const ZTerm* __restrict__ terms, const double* __restrict__ coefs,
                                                            float* __restrict__ out, int K, int npix, double scale,
                                                            double outside) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long npx = (long)npix * npix;
    if (idx >= npx) return;
    const int iy = (int)(idx / npix), ix = (int)(idx % npix);
    const double half = (npix - 1) / 2.0;
    const double x = ((double)ix - half) / half, y = ((double)iy - half) / half;
    const double rho = sqrt(x * x + y * y);
    const bool inside = !(rho > 1.0);
    double cx = 1.0, sx = 0.0;                        // cos(theta), sin(theta), theta = atan2(y, x)
    if (rho > 0.0) { cx = x / rho; sx = y / rho; }
    double pw[ZMAXN + 1], cm[ZMAXN + 1], sm[ZMAXN + 1];
    pw[0] = 1.0; cm[0] = 1.0; sm[0] = 0.0;
    for (int i = 1; i <= ZMAXN; ++i) {
        pw[i] = pw[i - 1] * rho;
        cm[i] = cm[i - 1] * cx - sm[i - 1] * sx;
        sm[i] = sm[i - 1] * cx + cm[i - 1] * sx;
    }
    for (int j = 0; j < K; ++j) {
        const ZTerm t = terms[j];
        double v = outside;
        if (inside) {
            double rad = 0.0;
            for (int k = 0; k < t.cnt; ++k) rad += coefs[t.off + k] * pw[t.n - 2 * k];
            const int am = t.m < 0 ? -t.m : t.m;
            v = t.norm * rad;
            if (t.m > 0) v *= cm[am];
            else if (t.m < 0) v *= sm[am];
        }
        out[(long)j * npx + idx] = (float)(v * scale);
    }
}

}  // namespace ppv

extern "C" {

// terms: K records {int n, int m, int off, int cnt, double norm} (24 bytes each, device); coefs: device doubles.
int ppv_zernike_basis(const void* terms, const double* coefs, float* out, int K, int npix, double scale, double outside,
                      hipStream_t stream) {
    if (!terms || !coefs || !out) return PPV_ERR_NULL;
    const long npx = (long)npix * npix;
    ppv::zernike_basis_kernel<<<(unsigned)((npx + 255) / 256), 256, 0, stream>>>((const ppv::ZTerm*)terms, coefs, out, K, npix,
                                                                               scale, outside);
    return ppv_last_error();
}

int ppv_zernike_max_order(void) { return ppv::ZMAXN; }

}  // extern "C"
