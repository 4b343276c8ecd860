"""Shared helpers of the parity tests."""
import numpy as np

C2 = dict(n=60000, half=50.0, step=0.2, P=12000, N=100)     # BASELINE config 2
C5 = dict(n=200000, half=100.0, step=0.2, P=30000, N=100)   # BASELINE config 5
C1 = dict(n=60000, half=50.0, step=1.0, P=12000, N=100)     # BASELINE config 1 (100x100)
REFDEF = dict(n=60000, half=60.0, step=0.2, P=24000, N=200)  # the reference's shipped config.py:46-61,119-120


def grid_args(half, step, z_min=-10.0, z_max=10.0):
    """(x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max, canvas_height)"""
    return (step, step, -half, -half, z_min, half, half, z_max, int(round(2 * half / step)))


def oracle_stage(O, pts32, P, N, half, step, order=0):
    return O.dataset_voxel_stage(pts32.astype(np.float64), P, N, *grid_args(half, step), order=order)


def pillars_by_cell(pillar, indices):
    """{(col,row): [9,N] block} of the occupied rows of a [9,P,N] tensor."""
    out = {}
    occ = np.nonzero(indices[:, 0])[0]
    for p in occ:
        out[(int(indices[p, 1]), int(indices[p, 2]))] = pillar[:, p, :]
    return out
