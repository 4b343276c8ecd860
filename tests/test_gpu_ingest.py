"""GPU parity of the lidar ingest pre-pass against the numpy
restatement of dataset.py:51-88, and of ingest -> voxelizer against the oracle's
create_pillars on the reference-style aggregated cloud."""
import numpy as np
import pytest

from util import grid_args

pytestmark = pytest.mark.gpu


def _rot(yaw, pitch):
    cz, sz, cy, sy = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    return np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])


def _transmat(rng):
    """ref_car_from_global . global_from_curr_car . curr_car_from_curr_sensor (dataset.py:52,71-76)
    for a car that moved a little between the sweep and the reference sample."""
    from pp_amd.ingest import transform_matrix
    t = rng.uniform(-1000, 1000, 3)
    yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-0.03, 0.03)
    global_from_car = transform_matrix(t, _rot(yaw, pitch))
    ref_from_global = transform_matrix(t + rng.uniform(-0.5, 0.5, 3),
                                       _rot(yaw + rng.uniform(-0.05, 0.05), pitch), inverse=True)
    car_from_sensor = transform_matrix([1.2, 0.0, 1.8], _rot(rng.uniform(-0.02, 0.02), 0.0))
    return ref_from_global @ global_from_car @ car_from_sensor


def test_ingest_matches_numpy_restatement_and_feeds_the_voxelizer(gpu, oracle):
    import torch
    from pp_amd import synth
    from pp_amd.ingest import LidarIngest
    from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig
    rng = np.random.default_rng(4)
    sweeps = []
    for s in range(3):                                       # num_sweeps aggregation
        raw = np.zeros((15000, 5), np.float32)               # x,y,z,intensity,ring
        raw[:, :4] = synth.lidar_like(15000, 20.0, s)
        raw[:, 4] = rng.integers(0, 64, 15000)
        mat = _transmat(rng)
        # 50 returns that land within min_dist of the reference car's origin: remove_close
        near = np.zeros((50, 4))
        near[:, :2] = rng.uniform(-4e-4, 4e-4, (50, 2))
        near[:, 2] = rng.uniform(-1, 1, 50)
        near[:, 3] = 1.0
        raw[100:150, :3] = (np.linalg.inv(mat) @ near.T).T[:, :3]
        sweeps.append((raw, mat))
    ing = LidarIngest(device=gpu)
    out = ing(sweeps)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = oracle.lidar_ingest(sweeps)                        # [n_kept,4] f64 (f32-valued)
    keep = ~np.isnan(got[:, 0])
    assert keep.sum() == len(ref) and (~keep).sum() >= 100
    # f64 4-term dot product rounded to f32 once: identical up to the summation order of the BLAS call
    assert np.abs(got[keep].astype(np.float64) - ref).max() <= 2e-6 * np.abs(ref[:, :3]).max()
    assert np.array_equal(got[keep][:, 3], ref[:, 3].astype(np.float32))
    # ingest output straight into the voxelizer == oracle on the reference-style cloud
    cfg = VoxelConfig.square(20.0, 0.2, 20000, 16)
    vox = PillarVoxelizer(cfg, device=gpu)
    pil, idx = vox(out)
    torch.cuda.synchronize()
    ref_p, ref_i, m = oracle.dataset_voxel_stage(got[keep].astype(np.float64), 20000, 16, *grid_args(20.0, 0.2))
    assert np.array_equal(idx[0].cpu().numpy(), ref_i) and np.array_equal(pil[0].cpu().numpy(), ref_p)


def test_ingest_argument_checks(gpu):
    import torch
    from pp_amd.ingest import LidarIngest
    ing = LidarIngest(device=gpu)
    with pytest.raises(ValueError):
        ing([(np.zeros((10, 3), np.float32), np.eye(4))])
    out = ing([(np.zeros((0, 5), np.float32), np.eye(4)), (np.ones((3, 4), np.float32), np.eye(4))])
    torch.cuda.synchronize()
    assert out.shape == (3, 4) and torch.equal(out.cpu(), torch.ones(3, 4))


def test_sweeps_in_one_launch_equal_the_per_sweep_calls(gpu):
    """pp_ingest_sweeps_dev (all sweeps of a sample per launch, sixteen at a time) against
    pp_ingest_dev sweep by sweep: the same bits, ragged sizes and an empty sweep included."""
    import ctypes
    import torch
    from pp_amd import _lib
    from pp_amd.ingest import LidarIngest
    rng = np.random.default_rng(11)
    sizes = [int(v) for v in rng.integers(0, 5000, 19)]
    sizes[3] = 0
    sweeps = [(rng.normal(0, 25, (n, 5)).astype(np.float32), _transmat(rng)) for n in sizes]
    ing = LidarIngest(device=gpu, min_dist=1.5)
    got = ing(sweeps)                                        # 16 + 3 sweeps: two launches
    ref = torch.empty_like(got)
    stream = ctypes.c_void_p(torch.cuda.current_stream(ing.device).cuda_stream)
    off = 0
    for raw, mat in sweeps:
        r = torch.from_numpy(raw).to(gpu)
        m = np.ascontiguousarray(mat, np.float64)
        rc = _lib.lib().pp_ingest_dev(ing._ctx.handle, stream, ctypes.c_void_p(r.data_ptr()), len(raw), 5,
                                      m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1.5,
                                      ctypes.c_void_p(ref.data_ptr() + off * 16))
        _lib.check(rc, "pp_ingest_dev")
        off += len(raw)
    torch.cuda.synchronize()
    a, b = got.cpu().numpy(), ref.cpu().numpy()
    assert off == len(a) == sum(sizes)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))      # NaN-poisoned rows included
    assert np.isnan(a[:, 0]).sum() > 0
