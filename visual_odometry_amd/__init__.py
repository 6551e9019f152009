"""visual_odometry_amd -- MI355X (gfx950) implementation of the data-parallel front end of
Heusini/visual_odometry: SIFT descriptor matching (k=2 + ratio test) and RANSAC hypothesis
generation/scoring for the 2D-2D (fundamental/essential matrix) and 3D-2D (P3P PnP) steps, behind
the reference's own Python call signatures.  Hand-written HIP kernels in ``csrc/`` behind the C ABI
of ``include/vo_hip.h``; no CPU fallback.
"""
__version__ = "0.1.0"
