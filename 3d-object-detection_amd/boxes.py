"""Anchor grid and box geometry as flat arrays (no lyft ``Box`` objects).

Counterparts of ``make_anchor_boxes`` (/root/reference utils/box_utils.py:111-159),
``boxes_to_image_space`` (:19-32) and the anchor constants of config.py:64-116.
The lyft_dataset_sdk ``Box.bottom_corners`` layout is a third-party behaviour
that is absent from the image; it is restated from its published definition
(corners (+l/2,-w/2), (+l/2,+w/2), (-l/2,+w/2), (-l/2,-w/2) rotated by yaw about
z), recalled and not verifiable here.
"""
from dataclasses import dataclass

import numpy as np

# config.py:64-89,109: class sizes (metres) -> cell units at STEP; three size groups
_STEP = 0.2


def _cells(wlh):
    a = np.array(wlh, np.float64)
    a[:2] = a[:2] / _STEP
    return a


ANIMAL, BICYCLE, BUS, CAR = _cells([.5, 1, .5]), _cells([.75, 2, 1.5]), _cells([3, 12.5, 3.5]), _cells([2, 5, 1.75])
EMERGENCY, MOTORCYCLE, OTHER = _cells([2.5, 6.5, 2.5]), _cells([1, 2.5, 1.5]), _cells([2.75, 8.5, 3.5])
PEDESTRIAN, TRUCK = _cells([.75, .75, 1.75]), _cells([3, 10, 3.5])
SMALL = np.mean(np.stack((ANIMAL, BICYCLE, PEDESTRIAN, MOTORCYCLE)), axis=0)   # config.py:93-94
MED = CAR                                                                     # config.py:95
LARGE = np.mean(np.stack((BUS, EMERGENCY, TRUCK, OTHER)), axis=0)              # config.py:96-97


@dataclass(frozen=True)
class AnchorConfig:
    fm_height: int
    fm_width: int
    fm_scale: float = 0.5
    dims: tuple = (tuple(MED), tuple(MED))
    yaws_deg: tuple = (0.0, 90.0)
    zs: tuple = (0.75, 0.75)

    @staticmethod
    def reference_default():
        """config.py:55,58-59,109-116: 300x300 feature map, 6 anchors per cell."""
        return AnchorConfig(300, 300, 0.5,
                            tuple(tuple(d) for d in (SMALL, SMALL, MED, MED, LARGE, LARGE)),
                            (0.0, 90.0) * 3, (.5, .5, .75, .75, 1.0, 1.0))

    @property
    def per_cell(self):
        return len(self.dims)

    @property
    def num_anchors(self):
        return self.fm_height * self.fm_width * len(self.dims)


def bottom_corners_xy(centers, wlh, yaw):
    """xy of ``Box.bottom_corners()`` for yaw-only boxes -> [...,4,2], counter-
    clockwise in a y-up frame (box_utils.py:27,149 call sites)."""
    centers = np.asarray(centers, np.float64)
    wlh = np.asarray(wlh, np.float64)
    yaw = np.asarray(yaw, np.float64)
    w, l = wlh[..., 0], wlh[..., 1]
    lx = np.stack([l / 2, l / 2, -l / 2, -l / 2], -1)
    ly = np.stack([-w / 2, w / 2, w / 2, -w / 2], -1)
    c, s = np.cos(yaw)[..., None], np.sin(yaw)[..., None]
    x = c * lx - s * ly + centers[..., 0:1]
    y = s * lx + c * ly + centers[..., 1:2]
    return np.stack([x, y], -1)


def make_anchors(cfg: AnchorConfig):
    """box_utils.py:111-159 -> dict of flat f64 arrays, anchors ordered
    (y, x, d): index = (y*fm_width + x)*per_cell + d, matching
    ``cls.permute(0,2,3,1)`` in model/loss.py:31-36."""
    dims = np.asarray(cfg.dims, np.float64)
    nd = dims.shape[0]
    yy, xx, dd = np.meshgrid(np.arange(cfg.fm_height), np.arange(cfg.fm_width), np.arange(nd),
                             indexing="ij")
    d = dd.reshape(-1)
    centers = np.stack([(xx.reshape(-1) + 0.5) / cfg.fm_scale,       # box_utils.py:137
                        (yy.reshape(-1) + 0.5) / cfg.fm_scale,       # box_utils.py:138
                        np.asarray(cfg.zs, np.float64)[d]], -1)      # box_utils.py:140
    wlh = dims[d]
    yaw = np.deg2rad(np.asarray(cfg.yaws_deg, np.float64))[d]
    corners = bottom_corners_xy(centers, wlh, yaw)
    # box_utils.py:152-155: (x1,y1,x2,y2) rows of anchor_xy.pkl -- corners 1,3 for the
    # rotated anchors (yaw > 0), corners 2,0 for the others
    rot = (np.asarray(cfg.yaws_deg, np.float64)[d] > 0)[:, None]
    xy = np.where(rot, np.concatenate([corners[:, 1], corners[:, 3]], 1),
                  np.concatenate([corners[:, 2], corners[:, 0]], 1))
    return {"corners": np.ascontiguousarray(corners), "centers": np.ascontiguousarray(centers),
            "wlh": np.ascontiguousarray(wlh), "yaw": np.ascontiguousarray(yaw),
            "xy": np.ascontiguousarray(xy)}


def anchor_type_table(cfg: AnchorConfig):
    """[per_cell, 13] f64 table for the on-the-fly anchor grid (pp_assign_targets_grid_dev):
    per anchor type the rotated bottom-corner offsets from the centre (x0,y0,..,x3,y3),
    then w, l, h, yaw (radians), z.  ``offset + centre`` is the last operation of
    ``bottom_corners_xy``, so the device's corners equal ``make_anchors``' bit for bit."""
    dims = np.asarray(cfg.dims, np.float64)
    yaw = np.deg2rad(np.asarray(cfg.yaws_deg, np.float64))
    off = bottom_corners_xy(np.zeros((dims.shape[0], 3)), dims, yaw).reshape(-1, 8)
    return np.ascontiguousarray(np.concatenate(
        [off, dims, yaw[:, None], np.asarray(cfg.zs, np.float64)[:, None]], 1))


def boxes_to_image_space(centers, wlh, yaw, canvas_height):
    """box_utils.py:19-32: ground-truth centres/corners with y flipped into rows."""
    centers = np.array(centers, np.float64, copy=True)
    corners = bottom_corners_xy(centers, wlh, yaw)
    centers[..., 1] = (canvas_height - 1) - centers[..., 1]
    corners[..., 1] = (canvas_height - 1) - corners[..., 1]
    return centers, corners
