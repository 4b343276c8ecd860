"""Writes the bench clouds as raw f32 for tools/lab/vox_lab: gen_pts.py out.bin B n half"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pp_amd import synth  # noqa: E402

out, B, n, half = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
np.stack([synth.lidar_like(n, half, s) for s in range(B)]).astype(np.float32).tofile(out)
