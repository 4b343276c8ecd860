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


# --------------------------------------------------------------------------- hand-off soak of the target kernels
def soak_gt_sets(H, classes=9):
    """Box sets that differ in everything the last-workgroup tail reads from the other workgroups: the number of
    boxes (so the number of workgroups, list slots and ticket groups), which anchors are positive, which are forced,
    the classes.  One set is empty (no tail at all: the scratch of the call before must still be in order afterwards)
    and one holds a single box."""
    from pp_amd import synth
    sets = [synth.gt_boxes(40, H, 13), synth.gt_boxes(17, H, 14, size_wl=(7.0, 15.0)), synth.gt_boxes(33, H, 15),
            synth.gt_boxes(1, H, 16), synth.gt_boxes(40, H, 17, size_wl=(12.0, 30.0), margin=80.0),
            {k: v[:0] for k, v in synth.gt_boxes(1, H, 18).items()}]
    for g in sets:
        g["classes"] = (g["classes"] % classes).astype(np.int32)
    return sets


def soak_order(n_sets, calls, seed=0):
    """A call sequence in which every ordered pair (previous set, this set) occurs, never the same set twice in a
    row: a read of the previous call's bytes always has different bytes to find."""
    rng = np.random.default_rng(seed)
    seq = [0]
    pairs = [(a, b) for a in range(n_sets) for b in range(n_sets) if a != b]
    rng.shuffle(pairs)
    for a, b in pairs:
        if seq[-1] != a:
            seq.append(a)
        seq.append(b)
    while len(seq) < calls:
        k = int(rng.integers(0, n_sets))
        if k != seq[-1]:
            seq.append(k)
    return seq[:max(calls, len(seq))]


def oracle_targets_for(O, anchors, g, H, thresh, classes=9):
    from pp_amd import boxes
    c_img, k_img = boxes.boxes_to_image_space(g["centers"], g["wlh"], g["yaw"], H)
    ref_c, ref_r, _ = O.create_target(anchors["corners"], k_img, anchors["centers"], c_img, anchors["wlh"],
                                      anchors["yaw"], g["centers"], g["wlh"], g["yaw"], g["classes"], H,
                                      pos_thresh=thresh, num_classes=classes)
    return ref_c, ref_r


def check_targets(cls_t, reg_t, ref_c, ref_r, tol=1e-6):
    cls_t, reg_t = cls_t.cpu().numpy(), reg_t.cpu().numpy()
    assert np.array_equal(cls_t, ref_c.astype(np.float32)), "class targets differ"
    assert np.array_equal(reg_t[:, 0], ref_r[:, 0].astype(np.float32)), "positive flags differ"
    assert np.array_equal(reg_t[:, 8], ref_r[:, 8].astype(np.float32)), "orientation bits differ"
    assert np.abs(reg_t - ref_r.astype(np.float32)).max() <= tol


class SideTraffic:
    """Unrelated memory traffic on a second stream WHILE the kernels under test run (the guide's rule: a hand-off is
    tested under uneven load, not on an idle chip), plus a flush of L2 / the Infinity Cache on the test's own stream
    every few calls."""

    def __init__(self, device, mib=256):
        import torch
        self.torch = torch
        self.side = torch.cuda.Stream(device)
        self.junk = torch.zeros((mib << 18,), dtype=torch.float32, device=device)
        self.junk2 = torch.zeros((32 << 18,), dtype=torch.float32, device=device)

    def flush(self):
        self.junk.add_(1.0)                      # 2 x mib of traffic on the test's stream

    def poke(self):
        with self.torch.cuda.stream(self.side):  # 64 MiB on the side stream, not waited for: overlaps the next calls
            self.junk2.add_(1.0)

    def close(self):
        self.side.synchronize()
