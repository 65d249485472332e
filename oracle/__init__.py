"""CPU oracle for the Camera + ResNet-101 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``privacy-preserving-vision_amd/``).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.

The oracle is a restatement (torch-CPU / numpy ops, identical arithmetic and
dtype promotions) of the reference's learned-optics camera and ResNet-101
encoder; every function cites the reference file:line it follows.

Parity pin status
-----------------
* IC ``OpticsZernike`` / FD ``Camera`` (forward and coefficient gradient) / RAFT
  ``CorrBlock`` and the on-the-fly ``alt_corr`` restatement of ``AlternateCorrBlock`` +
  ``alt_cuda_corr`` / FAN / ``DecoderWithAttention`` (forward, loss, every gradient) /
  ``pytorch_ssim``: PINNED against golden vectors generated in the build container by
  importing the reference's own Python (``tests/golden/make_golden.py``; fixtures
  ``tests/golden/*.npz``).
* Third-party arithmetic absent from /root/reference: ``poppy`` 1.0.3
  ``zernike_basis`` and ``cv2.circle`` -- *parity unpinned* (no reference
  fixture stores a basis or a mask); the golden generator feeds the reference
  the oracle's own basis / Euclidean disk (``oracle/zernike.py``), and the
  reference itself treats the basis as a cached ``.npy`` input.
* ``Encoder`` (``Image_Caption/models.py:8-54``): PINNED since round 3 against
  ``tests/golden/encoder.npz`` = the reference's own ``models.Encoder`` class run on CPU
  in train mode (``make_golden.py encoder``: output, trainable set, key / parameter order,
  input gradient, parameter gradients, running statistics).  torchvision itself is absent:
  the stand-in registered for ``torchvision.models.resnet101`` is assembled from
  ``oracle/resnet.py``'s pieces in torchvision's child order, and that restatement is
  cross-checked against the independent ResNet-101 v1.5 of ``transformers``
  (``tests/test_oracle_trunk_pin.py``).  torchvision's own source and pretrained weights
  stay *unpinned* (absent offline).
"""
