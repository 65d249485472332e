"""Analytic checks of the oracle's Zernike generator (poppy is absent: parity unpinned)."""
import numpy as np

from oracle import zernike as oz


def test_noll_indices():
    assert [oz.noll_indices(j) for j in range(1, 12)] == [
        (0, 0), (1, 1), (1, -1), (2, 0), (2, -2), (2, 2), (3, -1), (3, 1), (3, -3), (3, 3), (4, 0)]
    assert oz.noll_indices(350)[0] == 25 and oz.noll_indices(351)[0] == 25 and oz.noll_indices(352)[0] == 26


def test_known_terms_and_orthonormality():
    n = 257
    z = oz.zernike_basis(21, n)
    x = (np.arange(n) - (n - 1) / 2) / ((n - 1) / 2)
    xx, yy = np.meshgrid(x, x)
    rho2 = xx ** 2 + yy ** 2
    ins = rho2 <= 1
    assert np.allclose(z[0][ins], 1) and np.all(z[:, ~ins] == 0)
    assert np.allclose(z[1][ins], 2 * xx[ins]) and np.allclose(z[2][ins], 2 * yy[ins])
    assert np.allclose(z[3][ins], np.sqrt(3) * (2 * rho2[ins] - 1))
    assert np.allclose(z[5][ins], np.sqrt(6) * (xx[ins] ** 2 - yy[ins] ** 2))
    gram = np.einsum("iyx,jyx->ij", z, z) / ins.sum()
    assert np.abs(gram - np.eye(21)).max() < 0.05     # pixelated disk
