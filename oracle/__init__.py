"""CPU oracle for the 360-saliency hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in numpy / torch-CPU, the arithmetic of the reference
(hsientzucheng/CP-360-Weakly-Supervised-Saliency) for the path
equi -> cube -> CubePad -> ResNet-50-cubic -> CAM -> ConvLSTM -> cube -> equi.
Every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported CPU baseline.  The
product package (``cp_360_weakly_supervised_saliency_amd``) never imports it and
has no CPU fallback: it raises when ``libcp360.so`` (the HIP library) is missing.

Parity pinning: the reference ships no tests or golden vectors,
so the oracle is pinned against outputs of the reference itself, imported in the
build container by ``tests/golden/make_golden.py`` and committed as fixtures under
``tests/golden/``.  One boundary stays UNPINNED: ``cv2.remap`` (OpenCV is a
third-party dependency absent from /root/reference and from this image; the README
pins it only in prose as cv2 3.4.2).  ``oracle.equi_to_cube.remap_linear`` restates
OpenCV's published INTER_LINEAR algorithm (5-bit fixed-point coordinates,
BORDER_CONSTANT 0); the sampling *grids* that feed it are pinned by goldens.
"""
