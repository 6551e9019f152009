"""Diagnostic (GPU box, a -DVO_KNN_EXP_STATS build via VO_HIP_LIB): how often the kNN sweep's gate opens."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visual_odometry_amd import ops, sweep, synth, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
for n, B in ((8192, 32), (32768, 8)):
    g = sweep.frame_pairs_device(list(range(B)), n, n, synth.K_MALAGA, synth.IMAGE_SIZE["malaga"], dev)
    flags = torch.zeros((2,), dtype=torch.int32, device=dev)
    qi8, qn, _ = ops.desc_to_i8(g["q"].reshape(B * n, 128), flags[0:1]); ti8, tn, _ = ops.desc_to_i8(g["t"].reshape(B * n, 128), flags[1:2])
    z = (ctypes.c_ulonglong * 4)()
    lib.vo_knn_stats(z, 1)
    ops.knn2(qi8.reshape(B, n, 128), qn.reshape(B, n), ti8.reshape(B, n, 128), tn.reshape(B, n), None); torch.cuda.synchronize()
    lib.vo_knn_stats(z, 0)
    print(f"n={n}: units {z[0]}, gate open {z[1]} ({z[1] / z[0]:.3f}), lanes over the bound per unit {z[2] / z[0]:.3f}", flush=True)
