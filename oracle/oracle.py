"""NumPy/ctypes front end of the CPU oracle.

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never from the product package.

PARITY STATUS: parity unpinned against the real reference build (Boost missing,
no reference fixtures; see pp_oracle.h).  Pinned by SURVEY.md 5.9/8c probe values
and analytic IoU known answers (tests/test_oracle_*.py).

Citations are relative to /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpp_oracle.so")

OK, ERR_INDEX, ERR_VALUE, ERR_WINDING, ERR_NOMEM = 0, -2, -3, -4, -5
ORDER_ROW_MAJOR, ORDER_SCRAMBLED, ORDER_HASH = 0, 1, 2


def build(force=False):
    """Compile oracle/pp_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pp_oracle.c")
    hdr = os.path.join(_HERE, "pp_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpp_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def pybind_module_path():
    import sysconfig
    return os.path.join(_HERE, "pillars_oracle" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False):
    """Compile oracle/oracle_module.cpp (the oracle's two functions behind the reference's pybind11 module surface,
    data/pillars.cpp:429-435) with the reference's own build line (install_mods.sh:8: g++ -O3 -Wall -shared -std=c++11
    -fPIC + the pybind11 includes; c++14 because pybind11 3 needs it), linked with the C oracle's object code."""
    import pybind11
    import sysconfig
    src = [os.path.join(_HERE, "oracle_module.cpp"), os.path.join(_HERE, "pp_oracle.c"), os.path.join(_HERE, "pp_oracle.h")]
    out = pybind_module_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in src):
        return out
    obj = os.path.join(_HERE, "pp_oracle.o")
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-Wall", "-std=c11", "-fPIC", "-ffp-contract=off", "-c",
                           src[1], "-o", obj])
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-Wall", "-shared", "-std=c++14", "-fPIC",
                           "-ffp-contract=off", "-fvisibility=hidden", "-I" + pybind11.get_include(),
                           "-I" + sysconfig.get_paths()["include"], "-I" + _HERE, src[0], obj, "-o", out + ".tmp", "-lm"])
    os.replace(out + ".tmp", out)
    return out


def faithful_module_path():
    import sysconfig
    return os.path.join(_HERE, "pillars_faithful" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_faithful(force=False):
    """Compile oracle/faithful_module.cpp -- the baseline-faithful CPU variant of BASELINE.md section 4 (bounds-checked
    pybind11 accessors, one heap node per point, two std::unordered_map keyed on the cell's doubles) -- with the
    reference's own build line (install_mods.sh:8: g++ -O3 -Wall -shared -std=c++11 -fPIC + the pybind11 includes;
    c++14 because pybind11 3 needs it).  The polygon arithmetic is linked in from pp_oracle.c."""
    import pybind11
    import sysconfig
    src = [os.path.join(_HERE, "faithful_module.cpp"), os.path.join(_HERE, "pp_oracle.c"), os.path.join(_HERE, "pp_oracle.h")]
    out = faithful_module_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in src):
        return out
    obj = os.path.join(_HERE, "pp_oracle.o")
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-Wall", "-std=c11", "-fPIC", "-ffp-contract=off", "-c",
                           src[1], "-o", obj])
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O3", "-Wall", "-shared", "-std=c++14", "-fPIC",
                           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + _HERE,
                           src[0], obj, "-o", out + ".tmp", "-lm"])
    os.replace(out + ".tmp", out)
    return out


_faithful = None


def faithful_module():
    """The module `pillars_faithful`: create_pillars / make_ious with the reference's signatures AND its cost
    structure (what bench.py times as cpu_baseline.faithful)."""
    global _faithful
    if _faithful is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("pillars_faithful", build_faithful())
        _faithful = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_faithful)
    return _faithful


_pymod = None


def pybind_module():
    """The module `pillars_oracle`: create_pillars / make_ious with the reference's positional pybind11 signatures,
    on the oracle's reference-style C loops (what bench.py times as the CPU baseline)."""
    global _pymod
    if _pymod is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("pillars_oracle", build_pybind())
        _pymod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_pymod)
    return _pymod


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, dbl, vp, ci = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int
        L.ppo_grid_dims.argtypes = [dbl] * 6 + [ctypes.POINTER(i64)] * 2
        L.ppo_grid_dims.restype = ci
        L.ppo_scramble_mult.argtypes = [i64]
        L.ppo_scramble_mult.restype = i64
        L.ppo_create_pillars.argtypes = (
            [vp, i64, i64, i64] + [vp] + [i64] * 6 + [vp] + [i64] * 4 + [ci, ci]
            + [dbl] * 9 + [ci, ctypes.POINTER(i64)])
        L.ppo_create_pillars.restype = ci
        L.ppo_cell_counts.argtypes = [vp, i64, i64, i64, vp, i64] + [dbl] * 9
        L.ppo_cell_counts.restype = i64
        L.ppo_iou_pair.argtypes = [vp, vp, ctypes.POINTER(ci)]
        L.ppo_iou_pair.restype = dbl
        L.ppo_make_ious.argtypes = ([vp] + [i64] * 4 + [vp] + [i64] * 4 + [vp, i64, i64]
                                    + [vp, i64, i64] + [vp, i64, i64])
        L.ppo_make_ious.restype = ci
        _lib = L
    return _lib


def _raise(rc, what):
    if rc == OK:
        return
    if rc == ERR_INDEX:
        raise IndexError(f"{what}: index out of range")  # pybind11 index_error
    if rc == ERR_WINDING:
        raise ValueError(f"{what}: IOU < 0 (wrong corner winding)")
    if rc == ERR_NOMEM:
        raise MemoryError(what)
    raise ValueError(f"{what}: invalid argument (rc={rc})")


def _f64(a, name, writable=False):
    a = np.asarray(a) if not isinstance(a, np.ndarray) else a
    if a.dtype != np.float64:
        if writable:
            # the reference silently writes into a forcecast temporary and the
            # results vanish; the oracle refuses instead.
            raise TypeError(f"{name} must be a float64 array")
        a = a.astype(np.float64)
    if writable and not a.flags.writeable:
        raise TypeError(f"{name} must be writable")
    return a


def grid_dims(x_step, y_step, x_min, y_min, x_max, y_max):
    nx, ny = ctypes.c_int64(), ctypes.c_int64()
    _raise(lib().ppo_grid_dims(x_step, y_step, x_min, y_min, x_max, y_max,
                               ctypes.byref(nx), ctypes.byref(ny)), "grid_dims")
    return nx.value, ny.value


def scramble_mult(ncells):
    return lib().ppo_scramble_mult(ncells)


def create_pillars(points, tensor, indices, max_points_per_pillar, max_pillars,
                   x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max,
                   canvas_height, order=ORDER_SCRAMBLED):
    """data/pillars.cpp:236-398 (positional signature of pillars.cpp:236-249).

    Returns the number of non-empty cells (the reference returns None)."""
    points = _f64(points, "points")
    tensor = _f64(tensor, "tensor", writable=True)
    indices = _f64(indices, "indices", writable=True)
    if points.ndim != 2 or tensor.ndim != 3 or indices.ndim != 2:
        raise IndexError("create_pillars: wrong number of dimensions")
    if points.shape[0] and points.shape[1] < 4:
        raise IndexError("create_pillars: points needs >= 4 columns")
    ncell = ctypes.c_int64()
    rc = lib().ppo_create_pillars(
        points.ctypes.data, points.shape[0], points.strides[0], points.strides[1],
        tensor.ctypes.data, *tensor.shape, *tensor.strides,
        indices.ctypes.data, *indices.shape, *indices.strides,
        int(max_points_per_pillar), int(max_pillars),
        float(x_step), float(y_step), float(x_min), float(y_min), float(z_min),
        float(x_max), float(y_max), float(z_max), float(canvas_height),
        int(order), ctypes.byref(ncell))
    _raise(rc, "create_pillars")
    return ncell.value


def cell_counts(points, x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max,
                canvas_height):
    """[M,3] int64 (canvas_x, canvas_y, count) of every non-empty cell, row-major."""
    points = _f64(points, "points")
    cap = max(1, points.shape[0])
    out = np.zeros((cap, 3), np.int64)
    m = lib().ppo_cell_counts(points.ctypes.data, points.shape[0], points.strides[0],
                              points.strides[1], out.ctypes.data, cap,
                              float(x_step), float(y_step), float(x_min), float(y_min),
                              float(z_min), float(x_max), float(y_max), float(z_max),
                              float(canvas_height))
    if m < 0:
        _raise(int(m), "cell_counts")
    return out[:m].copy()


def iou_pair(anchor, gt):
    """data/pillars.cpp:132-172 for one (anchor CCW, gt CW) quad pair."""
    a = np.ascontiguousarray(anchor, np.float64).reshape(8)
    g = np.ascontiguousarray(gt, np.float64).reshape(8)
    s = ctypes.c_int()
    v = lib().ppo_iou_pair(a.ctypes.data, g.ctypes.data, ctypes.byref(s))
    _raise(s.value, "iou")
    return v


def make_ious(a_corners, g_corners, a_centers, g_centers, ious):
    """data/pillars.cpp:400-427 (positional signature of pillars.cpp:400-404)."""
    a_corners = _f64(a_corners, "a_corners")
    g_corners = _f64(g_corners, "g_corners")
    a_centers = _f64(a_centers, "a_centers")
    g_centers = _f64(g_centers, "g_centers")
    ious = _f64(ious, "ious", writable=True)
    A, G = a_corners.shape[0], g_corners.shape[0]
    if (a_corners.shape[1:] != (4, 2) or g_corners.shape[1:] != (4, 2)
            or a_centers.shape[0] < A or g_centers.shape[0] < G
            or a_centers.shape[1] < 2 or g_centers.shape[1] < 2
            or ious.shape[0] < A or ious.shape[1] < G):
        raise IndexError("make_ious: shape mismatch")
    rc = lib().ppo_make_ious(
        a_corners.ctypes.data, A, *a_corners.strides,
        g_corners.ctypes.data, G, *g_corners.strides,
        a_centers.ctypes.data, *a_centers.strides,
        g_centers.ctypes.data, *g_centers.strides,
        ious.ctypes.data, *ious.strides)
    _raise(rc, "make_ious")


# --------------------------------------------------------------------------- #
# caller glue: data/dataset.py:88-106                                          #
# --------------------------------------------------------------------------- #

def dataset_voxel_stage(lidar_points, max_pillars, max_points, x_step, y_step,
                        x_min, y_min, z_min, x_max, y_max, z_max, canvas_height,
                        order=ORDER_SCRAMBLED, data_mean=None, create=None):
    """np.zeros + create_pillars + transpose + f32 cast (+ the optional data_mean), exactly
    the work of data/dataset.py:89-106.  Returns
    (pillar[9,P,N] float32, indices[P,3] int64, num_cells).  ``create``: a module-style
    ``create_pillars`` (positional signature of pillars.cpp:236-249, returns None) to call instead
    of this file's ctypes one -- bench.py's CPU legs pass ``pybind_module().create_pillars``."""
    pillar = np.zeros((max_pillars, max_points, 9))          # dataset.py:89
    indices = np.zeros((max_pillars, 3))                     # dataset.py:90
    if create is not None:
        create(lidar_points, pillar, indices, max_points, max_pillars, x_step, y_step, x_min, y_min, z_min,
               x_max, y_max, z_max, canvas_height)           # dataset.py:92-97
        m = None
    else:
        m = create_pillars(lidar_points, pillar, indices, max_points, max_pillars,
                           x_step, y_step, x_min, y_min, z_min, x_max, y_max, z_max,
                           canvas_height, order)             # dataset.py:92-97
    pillar = pillar.transpose([2, 0, 1])                     # dataset.py:99
    pillar = np.ascontiguousarray(pillar, dtype=np.float32)  # dataset.py:101 (.float())
    if data_mean is not None:                                # dataset.py:102-105, f32 - f32
        shape = pillar.shape
        pillar = (pillar.reshape(-1) - np.asarray(data_mean, np.float32).reshape(-1)).reshape(shape)
    indices = indices.astype(np.int64)                       # dataset.py:106 (.long())
    return pillar, indices, m


# --------------------------------------------------------------------------- #
# boxes: lyft_dataset_sdk Box.bottom_corners (absent third-party, RECALLED)    #
# --------------------------------------------------------------------------- #

def _one_box_bottom_corners_xy(cx, cy, w, l, yaw):
    """One yaw-only lyft Box: the SDK builds the corner template in the box frame
    (x along the length, y along the width), rotates it with the orientation matrix and adds
    the centre; bottom_corners() are template columns 2,3,7,6 = (+l/2,-w/2), (+l/2,+w/2),
    (-l/2,+w/2), (-l/2,-w/2).  Scalar arithmetic, one corner at a time (deliberately not the
    vectorised formula of the product's boxes.py: two derivations, one result)."""
    import math
    rot = ((math.cos(yaw), -math.sin(yaw)), (math.sin(yaw), math.cos(yaw)))   # Rz(yaw), xy block
    out = []
    for sx, sy in ((1.0, -1.0), (1.0, 1.0), (-1.0, 1.0), (-1.0, -1.0)):
        bx, by = sx * l / 2.0, sy * w / 2.0
        out.append((rot[0][0] * bx + rot[0][1] * by + cx, rot[1][0] * bx + rot[1][1] * by + cy))
    return out


def box_bottom_corners_xy(center, wlh, yaw):
    """xy of Box.bottom_corners() for yaw-only boxes -> [...,4,2], counter-clockwise in a y-up
    frame.  lyft_dataset_sdk is not in the image (call sites utils/box_utils.py:27,149): this is
    its published layout, recalled.  One independent pin exists in the reference itself:
    make_anchor_boxes takes corners [2],[0] of an unrotated and [1],[3] of a 90-degree anchor as
    (top-left, bottom-right) of the NMS rectangle (box_utils.py:152-155, evaluate.py:127-139),
    which is only a valid x1<x2, y1<y2 box for this corner order (tests/test_host_logic.py)."""
    center = np.asarray(center, np.float64)
    wlh = np.asarray(wlh, np.float64)
    yaw = np.asarray(yaw, np.float64)
    shape = np.broadcast_shapes(center.shape[:-1], wlh.shape[:-1], yaw.shape)
    c2 = np.broadcast_to(center, shape + (center.shape[-1],)).reshape(-1, center.shape[-1])
    w2 = np.broadcast_to(wlh, shape + (wlh.shape[-1],)).reshape(-1, wlh.shape[-1])
    y2 = np.broadcast_to(yaw, shape).reshape(-1)
    out = np.empty((c2.shape[0], 4, 2))
    for i in range(c2.shape[0]):
        out[i] = _one_box_bottom_corners_xy(float(c2[i, 0]), float(c2[i, 1]), float(w2[i, 0]),
                                            float(w2[i, 1]), float(y2[i]))
    return out.reshape(shape + (4, 2))


def boxes_to_image_space(centers, wlh, yaw, canvas_height):
    """utils/box_utils.py:19-32: per box, corners = bottom_corners xy, then BOTH the centre's
    and the corners' y become (CANVAS_HEIGHT - 1) - y (image rows)."""
    centers = np.array(centers, np.float64, copy=True).reshape(-1, 3)
    wlh = np.asarray(wlh, np.float64).reshape(-1, 3)
    yaw = np.asarray(yaw, np.float64).reshape(-1)
    corners = np.empty((centers.shape[0], 4, 2))
    for i in range(centers.shape[0]):                                     # box_utils.py:24
        bc = _one_box_bottom_corners_xy(centers[i, 0], centers[i, 1], wlh[i, 0], wlh[i, 1], yaw[i])
        for k in range(4):
            corners[i, k, 0] = bc[k][0]
            corners[i, k, 1] = (canvas_height - 1) - bc[k][1]             # box_utils.py:29
        centers[i, 1] = (canvas_height - 1) - centers[i, 1]               # box_utils.py:30
    return centers, corners


def make_anchor_boxes(fm_height, fm_width, fm_scale, anchor_dims, anchor_yaws_deg,
                      anchor_zs):
    """utils/box_utils.py:111-159, loop for loop: anchors ordered (y, x, d); returns
    corners[A,4,2], centers[A,3], wlh[A,3], yaw[A] (radians)."""
    import math
    nd = len(anchor_dims)
    A = fm_height * fm_width * nd
    corners, centers = np.empty((A, 4, 2)), np.empty((A, 3))
    wlh, yaw = np.empty((A, 3)), np.empty(A)
    i = 0
    for y in range(fm_height):                                            # box_utils.py:133
        for x in range(fm_width):                                         # :134
            for d in range(nd):                                           # :135
                xc, yc, zc = (x + 0.5) / fm_scale, (y + 0.5) / fm_scale, float(anchor_zs[d])  # :136-139
                w, l, h = (float(v) for v in anchor_dims[d])              # :140-142
                th = math.radians(float(anchor_yaws_deg[d]))              # :143-144 Quaternion(degrees=)
                corners[i] = _one_box_bottom_corners_xy(xc, yc, w, l, th)  # :149-150
                centers[i] = (xc, yc, zc)                                 # :151
                wlh[i] = (w, l, h)
                yaw[i] = th
                i += 1
    return corners, centers, wlh, yaw


def anchor_xy_rows(corners, anchor_yaws_deg):
    """utils/box_utils.py:152-155: the (x1,y1,x2,y2) rows of anchor_xy.pkl -- corners 1 and 3 of
    a rotated anchor (yaw > 0), corners 2 and 0 otherwise."""
    nd = len(anchor_yaws_deg)
    out = np.empty((corners.shape[0], 4))
    for i in range(corners.shape[0]):
        a, b = (1, 3) if anchor_yaws_deg[i % nd] > 0 else (2, 0)
        out[i] = (*corners[i, a], *corners[i, b])
    return out


# --------------------------------------------------------------------------- #
# targets: utils/box_utils.py:70-109 and 162-232                               #
# --------------------------------------------------------------------------- #

def make_target(a_center, a_wlh, a_yaw, g_center, g_wlh, g_yaw, canvas_height):
    """utils/box_utils.py:70-109 for one (anchor, ground-truth) pair.
    g_center is in canvas space (y not yet flipped), as in the reference."""
    ax, ay, az = a_center
    gx, gy, gz = g_center
    aw, al, ah = a_wlh
    gw, gl, gh = g_wlh
    ad = np.sqrt(aw ** 2 + al ** 2)
    at, gt = float(a_yaw), float(g_yaw)
    gy = (canvas_height - 1) - gy
    dx = (gx - ax) / ad
    dy = (gy - ay) / ad
    dz = (gz - az) / ah
    dw = np.log(gw / aw)
    dl = np.log(gl / al)
    dh = np.log(gh / ah)
    if gt <= np.pi and gt >= np.pi / 2:
        gt -= np.pi
    elif gt >= -np.pi and gt <= -np.pi / 2:
        gt += np.pi
    dt = np.sin(gt - at)
    if ((gt - at) <= np.pi and (gt - at) >= np.pi / 2) or \
            ((gt - at) >= -np.pi and (gt - at) <= -np.pi / 2):
        ort = 1
    else:
        ort = 0
    return [1, dx, dy, dz, dw, dl, dh, dt, ort]


def create_target(anchor_corners, gt_corners, anchor_centers, gt_centers_img,
                  anchor_wlh, anchor_yaw, gt_centers_canvas, gt_wlh, gt_yaw,
                  gt_classes, canvas_height, pos_thresh=0.6, num_classes=9, reg_dims=8):
    """utils/box_utils.py:162-232 on flat arrays instead of lyft Box lists.

    gt_corners / gt_centers_img are the image-space arrays of
    boxes_to_image_space; gt_centers_canvas the un-flipped Box.center values
    that make_target flips itself (box_utils.py:83)."""
    A, G = len(anchor_corners), len(gt_corners)
    ious = np.zeros((A, G))
    make_ious(anchor_corners, gt_corners, anchor_centers, gt_centers_img, ious)
    cls_targets = np.zeros((A, num_classes))
    reg_targets = np.zeros((A, reg_dims + 1))
    if G == 0:
        # the reference cannot get here (np.max over an empty axis raises, box_utils.py:193; its samples always
        # carry a box): the library's documented extension is "no box -> all-zero targets", in these shapes
        return cls_targets, reg_targets, ious
    gt_box_classes = np.asarray(gt_classes, np.int32)
    max_ious = np.max(ious, axis=1)
    arg_max_ious = np.argmax(ious, axis=1)
    pos_anchors = np.where(max_ious > pos_thresh)[0]
    pos_boxes = arg_max_ious[pos_anchors]
    iousT = ious.transpose([1, 0])
    top_anchor_for_box = np.argmax(iousT, axis=1)
    filter_inds = np.nonzero(top_anchor_for_box)
    top_anchor_for_box = top_anchor_for_box[filter_inds]
    cls_targets[pos_anchors, gt_box_classes[pos_boxes]] = 1
    cls_targets[top_anchor_for_box, :] = 0
    cls_targets[top_anchor_for_box, gt_box_classes[filter_inds]] = 1
    for i, anch in enumerate(pos_anchors):
        g = pos_boxes[i]
        reg_targets[anch, :] = make_target(anchor_centers[anch], anchor_wlh[anch],
                                           anchor_yaw[anch], gt_centers_canvas[g],
                                           gt_wlh[g], gt_yaw[g], canvas_height)
    for i, anch in enumerate(top_anchor_for_box):
        g = filter_inds[0][i]
        reg_targets[anch, :] = make_target(anchor_centers[anch], anchor_wlh[anch],
                                           anchor_yaw[anch], gt_centers_canvas[g],
                                           gt_wlh[g], gt_yaw[g], canvas_height)
    return cls_targets, reg_targets, ious


# --------------------------------------------------------------------------- #
# lidar ingest: data/dataset.py:51-88 (lyft_dataset_sdk LidarPointCloud, RECALLED)#
# --------------------------------------------------------------------------- #

def lidar_ingest(sweeps, min_dist=0.001):
    """np restatement of the sweep loop of PPDataset.__getitem__ (dataset.py:54-88):
    for each (raw[n,C] f32, transmat 4x4 f64): from_file keeps the first four
    columns as a float32 [4,n] array; transform() computes
    transmat.dot(vstack(points[:3], ones)) in f64 and stores it back as f32;
    remove_close() drops points with |x| < r and |y| < r; sweeps are hstacked.
    Returns the [n_kept, 4] f64 array create_pillars receives (dataset.py:88)."""
    agg = np.zeros((4, 0))
    for raw, mat in sweeps:
        pts = np.asarray(raw, np.float32)[:, :4].T.copy()                    # [4,n] f32
        mat = np.asarray(mat, np.float64).reshape(4, 4)
        pts[:3, :] = mat.dot(np.vstack((pts[:3, :], np.ones(pts.shape[1]))))[:3, :]
        x_filt = np.abs(pts[0, :]) < min_dist
        y_filt = np.abs(pts[1, :]) < min_dist
        pts = pts[:, np.logical_not(np.logical_and(x_filt, y_filt))]
        agg = np.hstack((agg, pts))
    return agg.transpose([1, 0])


# --------------------------------------------------------------------------- #
# inference post-processing: evaluate.py:33-139, 231-245                       #
# --------------------------------------------------------------------------- #

def anchor_xy(corners, yaws_deg_per_anchor):
    """utils/box_utils.py:152-155: the (x1,y1,x2,y2) rows of anchor_xy.pkl."""
    rot = (np.asarray(yaws_deg_per_anchor) > 0)[:, None]
    return np.where(rot, np.concatenate([corners[:, 1], corners[:, 3]], 1),
                    np.concatenate([corners[:, 2], corners[:, 0]], 1))


def nms(boxes, scores, thresh):
    """torchvision.ops.nms (absent third party; published CPU kernel restated):
    f32 boxes (x1,y1,x2,y2), decreasing score order, drop when IoU > thresh.
    Ties in score are broken by the lower index (stable sort)."""
    boxes = np.asarray(boxes, np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    suppressed = np.zeros(len(boxes), bool)
    keep = []
    for _i, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(boxes[i, 0], boxes[rest, 0])
        yy1 = np.maximum(boxes[i, 1], boxes[rest, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[rest, 2])
        yy2 = np.minimum(boxes[i, 3], boxes[rest, 3])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thresh)]] = True
    return np.array(keep, np.int64)


def postprocess(cls_tensor, reg_tensor, a_centers, a_wlh, a_yaw, a_xy, canvas_height, x_step, y_step,
                x_min, y_min, pos_thresh=0.5, nms_thresh=0.1, max_out=100, num_classes=9, reg_dims=8):
    """evaluate.py:231-245 + make_pred_boxes (:33-89) + move_box_to_car_space (:91-125) for one
    sample; cls_tensor [Ac*9,H,W], reg_tensor [Ac*8,H,W] float32.  Returns (boxes[K,9] f64 rows
    x,y,z,w,l,h,yaw,score,class in car space, kept anchor ids[K])."""
    cls = np.asarray(cls_tensor, np.float32).transpose(1, 2, 0).reshape(-1, num_classes)
    reg = np.asarray(reg_tensor, np.float32).transpose(1, 2, 0).reshape(-1, reg_dims).copy()
    cls = (np.float32(1) / (np.float32(1) + np.exp(-cls))).astype(np.float32)      # torch.sigmoid
    reg[:, 6] = np.tanh(reg[:, 6])
    scores, classes = cls.max(-1), cls.argmax(-1)
    pos = np.where(scores > np.float32(pos_thresh))[0]
    nb = np.asarray(a_xy, np.float64).astype(np.float32)[pos]                       # box_nms :134
    nb[:, 1] = np.float32(canvas_height - 1) - nb[:, 1]
    nb[:, 3] = np.float32(canvas_height - 1) - nb[:, 3]
    keep = nms(nb, scores[pos], nms_thresh)
    final = pos[keep[:max_out]]
    out = np.zeros((len(final), 9))
    for r, i in enumerate(final):                                                   # make_pred_boxes
        off = reg[i]
        diag = np.sqrt(a_wlh[i, 0] ** 2 + a_wlh[i, 1] ** 2)
        bx = a_centers[i, 0] + off[0] * diag
        by = a_centers[i, 1] + off[1] * diag
        bz = a_centers[i, 2] + off[2] * a_wlh[i, 2]
        bw = np.exp(off[3]) * a_wlh[i, 0]
        bl = np.exp(off[4]) * a_wlh[i, 1]
        bh = np.exp(off[5]) * a_wlh[i, 2]
        yaw = np.arcsin(off[6]) + a_yaw[i]
        y = (canvas_height - 1) - by                                                # move_box_to_car_space
        out[r] = [bx * x_step + x_min, y * y_step + y_min, bz, bw * y_step, bl * x_step, bh, yaw,
                  scores[i], classes[i]]
    return out, final
