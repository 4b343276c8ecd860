"""Synthetic inputs of the benchmark and parity tests.

There is no network and no Lyft data: clouds and boxes are generated with
``numpy.random.default_rng(seed)`` (PCG64); seed = global sweep id.
"""
import numpy as np


def lidar_like(n, half, seed, max_range=None):
    """A 64-beam spinning-lidar-like cloud, float32 ``[n,4]`` (x, y, z, intensity).

    Beam elevation is drawn from linspace(-25deg, +3deg, 64), azimuth U[0, 2pi).
    Downward beams hit the ground at r = 1.8/tan(-elev) (clipped to
    [1, 1.2*half], jittered by 1%) with z = -1.8 + 0.05 N(0,1); the others return
    at r = half*U(0,1), z = U(-1, 2).  ``max_range`` restricts r (parity clouds
    that stay below the pillar cap use 0.45*half).
    """
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-25.0, 3.0, 64))[rng.integers(0, 64, n)]
    az = rng.uniform(0.0, 2 * np.pi, n)
    down = elev < 0
    r_ground = np.clip(1.8 / np.tan(np.where(down, -elev, 1.0)), 1.0, 1.2 * half)
    r_ground = r_ground * (1.0 + 0.01 * rng.standard_normal(n))
    r_free = half * rng.uniform(0.0, 1.0, n)
    z_ground = -1.8 + 0.05 * rng.standard_normal(n)
    z_free = rng.uniform(-1.0, 2.0, n)
    r = np.where(down, r_ground, r_free)
    z = np.where(down, z_ground, z_free)
    if max_range is not None:
        r = np.where(r > max_range, max_range * rng.uniform(0.0, 1.0, n), r)
    inten = rng.uniform(0.0, 255.0, n)
    pts = np.stack([r * np.cos(az), r * np.sin(az), z, inten], -1)
    return pts.astype(np.float32)


def gt_boxes(num, canvas, seed, size_wl=(10.0, 24.0), margin=50.0):
    """Ground-truth boxes in canvas (cell) space: centres U[margin, canvas-margin]^2,
    size (w,l) cells, yaw U(-pi,pi), class U{0..8}.  Returns dict of arrays."""
    rng = np.random.default_rng(1000 + seed)
    lo, hi = margin, max(margin + 1.0, canvas - margin)
    centers = np.stack([rng.uniform(lo, hi, num), rng.uniform(lo, hi, num),
                        rng.uniform(0.25, 1.25, num)], -1)
    wlh = np.stack([np.full(num, size_wl[0]) * rng.uniform(0.85, 1.15, num),
                    np.full(num, size_wl[1]) * rng.uniform(0.85, 1.15, num),
                    rng.uniform(1.4, 2.0, num)], -1)
    yaw = rng.uniform(-np.pi, np.pi, num)
    classes = rng.integers(0, 9, num).astype(np.int32)
    return {"centers": centers, "wlh": wlh, "yaw": yaw, "classes": classes}
