#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic sweeps that
are already resident in HBM: HIP pillar voxelizer -> PPFeatureNet -> scatter ->
backbone -> detection head (inference forward, f32).  Workload = BASELINE
configs[1]: 60k-point lidar-like clouds, 500x500 BEV grid, P=12000, N=100, D=9,
random backbone weights.  Multi-GPU: one process per GPU, every rank owns its
own sweeps (seed = global sweep id), no data-path collective ("weak" scaling).

The voxelizer runs software-pipelined over consecutive steps (the reference's DataLoader prefetch,
train.py:120-121): ONE launch per step (k_step) does the split stage of batch i, the tile stage of batch
i-1, the order stage of batch i-2 and the emit stage of batch i-3 side by side, and the network consumes
batch i-3 -- every step does one batch's worth of every stage (`--three-launch`: three dependent launches per step instead).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant hand-written kernel: k_step (= the whole voxelizer), algorithmic bytes per
                launch / its mean duration, timed live with HIP events bound to the dispatch packets during
                the timed steps; `three_launch`: the same forward with k_split, k_tile, k_emit as three
                launches per step (their durations, their sum's fraction, k_emit's own)
  voxelizer_only   wall time per voxelizer call alone, pipelined and (`three_launch`) as three launches:
                default (scrambled) order, `row_major_order`, `one_sweep_per_launch` (configs[3]'s per-GPU
                shape), `c1_shapes` (configs[0]'s 100x100 grid on the GPU)
  fused_feature_net  SURVEY 8f rank 1 measured beside the headline, with its own roofline record
  train_c3      BASELINE configs[2] (and the RCCL leg of configs[3] when N > 1): HIP target
                assignment + loss forward/backward + gradient and loss-scalar all-reduces
  stress_c5     BASELINE configs[4] shapes on one GPU: 200k points, 1000x1000 grid, P=30000,
                voxelizer only, both paths, B sweeps and one sweep per launch
  reference_default  the reference's shipped config.py sizes (600x600, P=24000, N=200; 540 000 anchors)
  next_rows     SURVEY 8(f) ranks 3 and 2 in the driver's run: lidar ingest (5 sweeps of a sample in one launch) and the
                inference post-processing (scores, sort, NMS, decode of a batch), us per call and fraction for their bytes
  dropin_host   the pybind11 drop-in module's own speed on host arrays beside the CPU loop (PCIe inclusive), with meets_50x
  cpu_baseline  the CPU oracle's reference-style voxel stage (hash map of heap nodes + the
                caller's np.zeros/transpose/.float() glue, data/dataset.py:89-106) timed on
                this host, 1 core, at configs[1]'s grid and (`c1`) at configs[0]'s 100x100 grid
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import pp_amd  # noqa: E402
from pp_amd import _lib, shard, synth  # noqa: E402
from pp_amd.pipeline import PillarPipeline  # noqa: E402
from pp_amd.voxelizer import PillarVoxelizer, VoxelConfig  # noqa: E402

METRIC = "lidar sweeps/sec end-to-end fwd (pillarize+backbone), 60k pts, 500×500 BEV"
HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md:36
HBM_COPY_MEASURED = 6.29e12  # B/s, the same line: "6.29 TB/s measured (float4 copy, 79%)"
N_POINTS, HALF, STEP, P, N = 60000, 50.0, 0.2, 12000, 100
C5 = dict(n=200000, half=100.0, step=0.2, P=30000, N=100)   # BASELINE configs[4] shapes
C1 = dict(n=60000, half=50.0, step=1.0, P=12000, N=100)     # BASELINE configs[0]: 100x100 grid


_CPU_WORKER = r"""
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
import pp_amd
from pp_amd import synth
from oracle import oracle as O
n, half, step, P, N, budget = int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
variant = sys.argv[8] if len(sys.argv) > 8 else "port"
pts = synth.lidar_like(n, half, 0).astype(np.float64)
args = (P, N, step, step, -half, -half, -10.0, half, half, 10.0, int(2 * half / step))
# the CPU loop behind the reference's pybind11 signatures: the plain-C port, or the baseline-faithful variant
cp = (O.faithful_module() if variant == "faithful" else O.pybind_module()).create_pillars
for _ in range(2):
    O.dataset_voxel_stage(pts, *args, create=cp)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < budget:
    O.dataset_voxel_stage(pts, *args, create=cp); k += 1
print(k / (time.perf_counter() - t0))
"""


def _cpu_module(variant):
    from oracle import oracle as O
    return O.faithful_module() if variant == "faithful" else O.pybind_module()


def _cpu_voxel_stage(n, half, step, p, nn, seconds_budget, variant="port"):
    """median seconds per call of the CPU voxel stage on one core (`variant`: the plain-C port of oracle/pp_oracle.c, or
    oracle/faithful_module.cpp with the reference's bounds-checked accessors and node-based maps)"""
    from oracle import oracle as O
    pts = synth.lidar_like(n, half, 0).astype(np.float64)  # dataset.py:82 hands over f64
    args = (p, nn, step, step, -half, -half, -10.0, half, half, 10.0, int(round(2 * half / step)))
    cp = _cpu_module(variant).create_pillars   # "through the same pybind11 signatures"
    for _ in range(2):
        O.dataset_voxel_stage(pts, *args, create=cp)
    times = []
    t_end = time.perf_counter() + seconds_budget
    while time.perf_counter() < t_end or len(times) < 5:
        t0 = time.perf_counter()
        O.dataset_voxel_stage(pts, *args, create=cp)
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times), float(np.min(times)), float(np.max(times))


def _cpu_processes(count, budget, variant):
    """`count` fresh single-threaded CPU processes of the same voxel stage side by side (never a fork of this
    GPU-initialised process): total sweeps/s"""
    import subprocess
    cmd = [sys.executable, "-c", _CPU_WORKER, ROOT, str(N_POINTS), str(HALF), str(STEP), str(P), str(N), str(budget), variant]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(count)]
    rates = [float(p.communicate(timeout=budget + 240)[0].strip().splitlines()[-1]) for p in procs]
    return float(sum(rates))


def _cpu_variant_record(variant, seconds_budget, workers, worker_budget, all_budget):
    """one CPU variant at 1 core, `workers` processes (DataLoader(num_workers=4), config.py:139) and all host cores
    (1 core, num_workers, all cores; capped at 64 processes)"""
    med, calls, lo, hi = _cpu_voxel_stage(N_POINTS, HALF, STEP, P, N, seconds_budget, variant)
    rec = {"one_core": {"value": 1.0 / med, "unit": "sweeps/s", "cores": 1, "calls": calls,
                        "ms_min_median_max": [lo * 1e3, med * 1e3, hi * 1e3]}}
    n_all = max(1, min(os.cpu_count() or 1, 64))
    for key, count, budget in (("workers", workers, worker_budget), ("all_cores", n_all, all_budget)):
        try:
            rec[key] = {"processes": count, "value": _cpu_processes(count, budget, variant), "unit": "sweeps/s"}
        except Exception as e:  # the single-core figure stands on its own
            rec[key] = {"processes": count, "value": None, "error": str(e)[:200]}
    rec["all_cores"]["host_cores"] = os.cpu_count()
    return rec, (med, calls, lo, hi)


def cpu_baseline(seconds_budget=8.0, workers=4, worker_budget=4.0, c1_budget=4.0, all_budget=4.0):
    """The CPU voxel stage beside the GPU path, in BOTH CPU variants:
      port      oracle/pp_oracle.c's reference-style loop (hash map of heap nodes, plain loads) -- the faster one, so the
                conservative one for any GPU/CPU ratio: the top-level `value` stays this one;
      faithful  oracle/faithful_module.cpp: what BASELINE.md section 4 describes (pybind11 .at() / .mutable_at() on every
                element, a heap node per point, two std::unordered_map keyed on the cell's doubles), pinned bit for bit
                against the oracle in tests/test_oracle_pillars.py.
    Each on one core, as `workers` processes side by side like the reference's DataLoader (num_workers = 4,
    config.py:139) and with every host core busy."""
    from oracle import oracle as O
    O.build()
    O.build_pybind()
    O.build_faithful()
    port, (med, calls, lo, hi) = _cpu_variant_record("port", seconds_budget, workers, worker_budget, all_budget)
    faithful, (fmed, fcalls, flo, fhi) = _cpu_variant_record("faithful", seconds_budget, workers, worker_budget, all_budget)
    out = {"value": 1.0 / med, "unit": "sweeps/s", "cores": 1, "kind": "port", "variant": "port",
           "binding": "pybind11 modules `pillars_oracle` (oracle/oracle_module.cpp, the C port) and `pillars_faithful` "
                      "(oracle/faithful_module.cpp): the reference's positional signatures, array_t<double> arguments, built "
                      "with the reference's flags (-O3 -fPIC, install_mods.sh:8)",
           "sample": f"{calls} calls of the voxel stage only (np.zeros + create_pillars "
                     f"[reference-style hash map of heap nodes] + transpose + f32 cast, "
                     f"dataset.py:89-106) on one {N_POINTS}-pt cloud, median {med * 1e3:.1f} ms "
                     f"(min {lo * 1e3:.1f}, max {hi * 1e3:.1f}; the median moves by +-25 % between boxes of "
                     f"this pool); host has {os.cpu_count()} cores; the backbone is not part of this leg; "
                     f"`faithful`: the same stage with bounds-checked accessors and node-based maps, {fcalls} calls, median "
                     f"{fmed * 1e3:.1f} ms (min {flo * 1e3:.1f}, max {fhi * 1e3:.1f})",
           "ms_min_median_max": [lo * 1e3, med * 1e3, hi * 1e3],
           "port": port, "faithful": faithful,
           # (the round-5 keys, kept for the readers of earlier lines)
           "workers": dict(port["workers"], what=f"{workers} concurrent CPU processes of the port's voxel stage, like "
                                                 f"DataLoader(num_workers={workers})"),
           "all_cores": dict(port["all_cores"], what="min(host cores, 64) concurrent single-threaded processes of the port")}
    # BASELINE configs[0]: the same cloud on the 100x100 grid (1 m cells), CPU path only
    c1 = {}
    for variant in ("port", "faithful"):
        med1, calls1, lo1, hi1 = _cpu_voxel_stage(C1["n"], C1["half"], C1["step"], C1["P"], C1["N"], c1_budget / 2, variant)
        c1[variant] = {"value": 1.0 / med1, "unit": "sweeps/s", "cores": 1, "calls": calls1,
                       "ms_min_median_max": [lo1 * 1e3, med1 * 1e3, hi1 * 1e3]}
    out["c1"] = dict(c1["port"], kind="port", faithful=c1["faithful"],
                     sample=f"configs[0]: one {C1['n']}-pt cloud, 100x100 grid (step {C1['step']} m), P={C1['P']} "
                            f"N={C1['N']}; port median {c1['port']['ms_min_median_max'][1]:.1f} ms, faithful "
                            f"{c1['faithful']['ms_min_median_max'][1]:.1f} ms")
    return out


def _median_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3


def dropin_record(gpu_ms, cpu_ms, what, faithful_ms=None):
    """one leg of `dropin_host`: the module call on host arrays (PCIe inclusive) beside the CPU figures -- `cpu_ms` /
    `speedup` / `meets_50x` against the plain-C port (the faster CPU variant: conservative), `faithful_*` against the
    baseline-faithful variant of BASELINE.md section 4 (oracle/faithful_module.cpp)"""
    ratio = cpu_ms / gpu_ms
    rec = {"hip_ms": gpu_ms, "cpu_ms": cpu_ms, "speedup": ratio, "meets_50x": bool(ratio >= 50.0), "what": what}
    if faithful_ms is not None:
        rec.update(faithful_cpu_ms=faithful_ms, faithful_speedup=faithful_ms / gpu_ms,
                   faithful_meets_50x=bool(faithful_ms / gpu_ms >= 50.0))
    return rec


def dropin_host(reps=15):
    """The drop-in surface's OWN speed (north_star: ">= 50x the CPU pybind11 pillarizer ... at matched output", "data/
    dataset.py calls the same symbols"): the built pybind11 module `pillars` (native/pillars*.so, what
    install_mods.sh:8-10 would move into data/) called the way data/dataset.py:88-106 and utils/box_utils.py:181-183
    call it -- float64 NumPy arrays on the HOST, outputs mutated in place, PCIe transfers and the host-side scatter
    included -- beside the CPU oracle's reference-style functions on the same arrays through the same kind of module
    (`pillars_oracle`, pybind11; 1 core; a C port without the reference's bounds-checked accessors, faster per call
    than the reference's Boost build as the survey probed it: conservative for the ratio)."""
    import importlib.util
    from oracle import oracle as O
    from pp_amd import boxes
    O.build()
    spec = importlib.util.spec_from_file_location("pillars", _lib.build_pybind_module())
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pts = synth.lidar_like(N_POINTS, HALF, 0).astype(np.float64)          # dataset.py:82 hands over f64
    agg = np.ascontiguousarray(pts.T)                                      # [4, n]; dataset.py:88 passes its transpose view
    cp_args = (N, P, STEP, STEP, -HALF, -HALF, -10.0, HALF, HALF, 10.0, int(round(2 * HALF / STEP)))
    T, I = np.zeros((P, N, 9)), np.zeros((P, 3))

    def call_hip():
        mod.create_pillars(agg.transpose([1, 0]), T, I, *cp_args)

    cpu_mod = O.pybind_module()               # the oracle behind the same pybind11 signatures
    fth_mod = O.faithful_module()             # ... and the baseline-faithful variant (bounds-checked accessors, node maps)

    def call_cpu():
        cpu_mod.create_pillars(agg.transpose([1, 0]), T, I, *cp_args)

    def call_fth():
        fth_mod.create_pillars(agg.transpose([1, 0]), T, I, *cp_args)

    def glue(create):      # data/dataset.py:88-106 around the call, statement for statement
        def run():
            pillar = np.zeros((P, N, 9))
            indices = np.zeros((P, 3))
            create(agg.transpose([1, 0]), pillar, indices, *cp_args)
            pillar = pillar.transpose([2, 0, 1])
            pillar = torch.from_numpy(pillar).float()
            indices = torch.from_numpy(indices).long()
            return pillar, indices
        return run
    hip_call, _ = _median_ms(call_hip, 2 * reps, warm=4)
    cpu_call, _ = _median_ms(call_cpu, max(5, reps // 2))
    fth_call, _ = _median_ms(call_fth, max(5, reps // 2))

    def fresh(create):     # the call on arrays FRESH from np.zeros (data/dataset.py:89-90): their first touch -- a page fault
        ts = []            # per 4 KB of every row written -- falls inside the call, whoever makes it
        for _ in range(2 + max(7, reps // 2)):
            pillar, indices = np.zeros((P, N, 9)), np.zeros((P, 3))
            t0_ = time.perf_counter()
            create(agg.transpose([1, 0]), pillar, indices, *cp_args)
            ts.append(time.perf_counter() - t0_)
            del pillar, indices
        return float(np.median(ts[2:])) * 1e3
    hip_fresh, cpu_fresh = fresh(mod.create_pillars), fresh(cpu_mod.create_pillars)
    # the caller's own statements are page-fault bound (np.zeros of 86 MB per run): the three variants take turns, so that
    # whatever the allocator and the page cache do drifts over all of them alike
    glues = [glue(mod.create_pillars), glue(cpu_mod.create_pillars), glue(fth_mod.create_pillars)]
    gt_ = [[], [], []]
    for rep_ in range(2 + max(7, reps // 2)):
        for k_, fn_ in enumerate(glues):
            t0_ = time.perf_counter()
            fn_()
            if rep_ >= 2:
                gt_[k_].append(time.perf_counter() - t0_)
    hip_glue, cpu_glue, fth_glue = (float(np.median(v)) * 1e3 for v in gt_)
    anchors = boxes.make_anchors(boxes.AnchorConfig(250, 250))
    gt = synth.gt_boxes(40, 500, 0)
    c_img, k_img = boxes.boxes_to_image_space(gt["centers"], gt["wlh"], gt["yaw"], 500)
    ious = np.zeros((anchors["corners"].shape[0], 40))
    hip_iou, _ = _median_ms(lambda: mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious), reps)
    flip = [0]

    def iou_changed():     # one anchor value edited in place before every call: the gather finds it, the whole set goes up again
        flip[0] ^= 1
        anchors["corners"][7, 0, 0] += 1e-3 if flip[0] else -1e-3
        mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious)
    hip_iou_changed, _ = _median_ms(iou_changed, reps)
    if flip[0]:
        anchors["corners"][7, 0, 0] -= 1e-3
    cpu_iou, _ = _median_ms(lambda: cpu_mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious), 5, warm=1)
    fth_iou, _ = _median_ms(lambda: fth_mod.make_ious(anchors["corners"], k_img, anchors["centers"], c_img, ious), 3, warm=1)
    return {
        "module": "native/" + os.path.basename(_lib.pybind_module_path()) + " (pybind11, csrc/pillars_module.cpp -> "
                  "pp_create_pillars_f64 / pp_make_ious_f64)",
        "create_pillars_call": dict(dropin_record(
            hip_call, cpu_call, "pillars.create_pillars(points f64 [n,4] strided view, tensor f64 [P,N,9], indices f64 "
            "[P,3], ...) at configs[1]'s shapes on pre-zeroed arrays, the call alone; median of "
            f"{2 * reps}; cpu = the oracle's reference-style create_pillars (hash map of heap nodes) on one core; the module "
            "gathers, scatters and (make_ious) zero-fills on a pool of host threads (PP_HOST_THREADS, default 8)", fth_call),
            fresh_arrays={"hip_ms": round(hip_fresh, 4), "cpu_ms": round(cpu_fresh, 4),
                          "speedup": round(cpu_fresh / hip_fresh, 2),
                          "what": "the same call on arrays fresh from np.zeros, as data/dataset.py:89-90 hands them over: "
                                  "the first-touch page faults of the rows written are inside the call on either side"}),
        "create_pillars_in_dataset_glue": dropin_record(
            hip_glue, cpu_glue, "the same call inside the reference caller's own statements (data/dataset.py:88-106: two "
            "np.zeros incl. the 86 MB f64 tensor, the call, transpose to [9,P,N], .float(), .long()): the glue is "
            "reference code and costs the same on both sides", fth_glue),
        "make_ious_call": dict(dropin_record(
            hip_iou, cpu_iou, "pillars.make_ious(a_corners [A,4,2], g_corners, a_centers, g_centers, ious [A,G]) at "
            "A=125000, G=40 (configs[2]), the SAME anchor arrays call after call as utils/box_utils.py:181-183 hands them "
            "in: the gather compares them with the set resident on the device (11 MB, uploaded only when a bit changed), "
            "the ~8 000 entries that are not zero come back as 16-byte records, the caller's 40 MB matrix is zero-filled "
            "by host threads while the kernel runs", fth_iou),
            anchors_changed_every_call_ms=round(hip_iou_changed, 4),
            anchors_changed_every_call_speedup=round(cpu_iou / hip_iou_changed, 2)),
        "note": "PCIe-inclusive compatibility numbers, never `value`.  `speedup` is against the plain-C port (the faster "
                "CPU variant), `faithful_speedup` against the variant with the reference's bounds-checked accessors and "
                "node-based maps.  Inside the caller's own statements the reference's host-side work bounds the ratio: "
                "np.zeros + transpose + f32 cast of an 86 MB f64 tensor; for make_ious the 40 MB matrix the signature "
                "demands is zero-filled on the host"}


def next_rows_record(pipe, points, dev):
    """SURVEY 8(f) rows either side of the path, measured in the driver's run (their parity tests are tests/test_gpu_ingest.py
    and tests/test_gpu_postprocess.py): the lidar ingest in front of the voxelizer (rank 3, data/dataset.py:51-88) and the
    inference post-processing behind the head (rank 2, evaluate.py:231-245).  Both are small launches: the fraction of the
    HBM roofline for their own bytes is reported for the record, what bounds them is latency."""
    from pp_amd import boxes
    from pp_amd.ingest import LidarIngest, transform_matrix
    from pp_amd.postprocess import Detector
    rng = np.random.default_rng(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        for _ in range(5):
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    # ingest: one sample = 5 sweeps of 60 000 raw rows (x, y, z, intensity, ring: 20 B) -> [300000, 4] f32 (16 B per point)
    ns = 5
    raw = [torch.from_numpy(rng.normal(0, 30, (N_POINTS, 5)).astype(np.float32)).to(dev) for _ in range(ns)]
    th = 0.3
    mat = transform_matrix([1.0, 2.0, 0.5], [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    ing = LidarIngest(device=dev)
    sweeps = [(r, mat) for r in raw]
    ing_us = timed(lambda: ing(sweeps), 200)
    ing_bytes = ns * N_POINTS * (20 + 16)
    # post-processing: the head's outputs of one forward (B sweeps), logits shifted so that ~1 % of the anchors pass the
    # score threshold (evaluate.py's regime after training; random weights alone pass none or all)
    with torch.no_grad():
        cls, reg = pipe.forward(points)
        cls, reg = cls.clone(), reg.clone()
    acfg = pipe.anchor_cfg
    det = Detector(boxes.make_anchors(acfg), acfg, pipe.vox_cfg.canvas_height, STEP, STEP, -HALF, -HALF,
                   pos_thresh=0.2, nms_thresh=0.1, device=dev)
    score = torch.sigmoid(cls.float()).reshape(cls.shape[0], acfg.per_cell, 9, -1).amax(2)
    q = float(torch.quantile(score.flatten()[::7].float(), 0.99))
    shift = float(np.log(0.2 / 0.8) - np.log(q / (1.0 - q)))          # moves the 99th percentile score to the threshold
    cls_s = cls + shift
    cand = [(torch.sigmoid(cls_s[i].float()).reshape(acfg.per_cell, 9, -1).amax(1) > 0.2).sum().item()
            for i in range(cls.shape[0])]
    dec_us = timed(lambda: det(cls_s, reg), 50)
    kept = [int(c) for c in det(cls_s, reg)[2].tolist()]
    dec_bytes = (cls.numel() + reg.numel()) * 4
    return {
        "ingest": {"kernel": "pp::k_ingest_sweeps (all sweeps of a sample in one launch)", "us_per_sample": ing_us,
                   "sweeps": ns, "points_per_sweep": N_POINTS, "bytes_per_launch": ing_bytes,
                   "bytes_what": "20 B raw row in + 16 B point out, per point", "frac": ing_bytes / (ing_us * 1e-6) / HBM_PEAK},
        "postprocess": {"kernel": "pp::k_score + pp::k_sort_runs + pp::k_nms (three launches per batch)",
                        "us_per_batch": dec_us, "us_per_sample": dec_us / cls.shape[0], "samples": int(cls.shape[0]),
                        "candidates_per_sample": [int(c) for c in cand], "kept_per_sample": kept,
                        "bytes_per_launch": dec_bytes, "bytes_what": "the head's cls + reg maps read once",
                        "frac": dec_bytes / (dec_us * 1e-6) / HBM_PEAK},
        "what": "SURVEY 8(f) rank 3 (ingest) and rank 2 (post-processing), timed with events around back-to-back calls"}


def kernel_means_us(vox):
    """mean duration (us) of the voxelizer's kernels over the calls since set_timing: the three kernels of the
    plain calls, or k_step alone (recorded in the k_emit column) after pipelined calls"""
    ms = {k: vox.read_kernel_ms(w) for k, w in
          (("k_split", _lib.KERNEL_SPLIT), ("k_tile", _lib.KERNEL_TILE))}
    emit = vox.read_kernel_ms(_lib.KERNEL_EMIT)      # last: empties the ring
    if not ms["k_split"]:
        return {"k_step": float(np.mean(emit)) * 1e3 if emit else float("nan")}, len(emit)
    ms["k_emit"] = emit
    return {k: (float(np.mean(v)) * 1e3 if v else float("nan")) for k, v in ms.items()}, len(emit)


def three_launch_record(kern_us, bytes_per_launch):
    total_s = sum(kern_us.values()) * 1e-6
    return {"kernels_us": kern_us, "sum_us": total_s * 1e6,
            "pipeline_frac": bytes_per_launch / total_s / HBM_PEAK,
            "k_emit_frac": bytes_per_launch / (kern_us["k_emit"] * 1e-6) / HBM_PEAK,
            "what": "pp_voxelize_dev: k_split, k_tile, k_emit as three dependent launches of ONE batch; "
                    "pipeline_frac = the same algorithmic bytes over the SUM of the three"}


def roofline_record(kern_us, launches, bytes_per_launch, traffic=None, traffic_source=None, three=None,
                    step_kernel="pp::k_step"):
    """the dominant hand-written kernel against the HBM roofline.  Pipelined calls: ONE kernel, k_step, does a
    batch's worth of all three stages per launch, so its fraction IS the whole voxelizer's.  Plain calls:
    k_emit (the dense store), with the sum of the three kernels beside it."""
    if "k_step" in kern_us:
        dur_us, name = kern_us["k_step"], (step_kernel + " (one launch = the split stage of batch i, the tile "
                                           "stage of batch i-1, the order stage of batch i-2 and the emit stage of "
                                           "batch i-3: one batch's worth of every voxelizer stage)")
    else:
        dur_us, name = kern_us["k_emit"], "pp::k_emit<float,0>"
    achieved = bytes_per_launch / (dur_us * 1e-6) / 1e9
    rec = {"bound": "hbm", "kernel": name, "achieved": achieved,
           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved * 1e9 / HBM_PEAK,
           # SURVEY 8(d) asks for both: the 8.0 TB/s specification (`peak`, `frac`) and the guide's measured
           # float4-copy rate (MI355X_MICROARCH.md: 6.29 TB/s, 79 % of the specification)
           "peak_measured_copy": HBM_COPY_MEASURED / 1e9, "frac_of_measured_copy": achieved * 1e9 / HBM_COPY_MEASURED,
           "traffic": traffic, "traffic_source": traffic_source,
           "bytes_per_launch": bytes_per_launch, "avg_launch_us": dur_us, "launches_timed": launches}
    if "k_step" in kern_us:
        rec["pipeline_frac"] = rec["frac"]       # the kernel is the whole voxelizer
        if three is not None:
            rec["three_launch"] = three
    else:
        t = three_launch_record(kern_us, bytes_per_launch)
        rec["pipeline"] = {"kernels_us": kern_us, "sum_us": t["sum_us"],
                           "achieved": bytes_per_launch / (t["sum_us"] * 1e-6) / 1e9, "what": t["what"]}
        rec["pipeline_frac"] = t["pipeline_frac"]
    return rec


def rotating_outputs(batch, p, n, dev, total_bytes=512 << 20, most=12):
    """Output buffers for a voxelizer-only loop, used in turn, together at least twice the 256 MB Infinity Cache: a
    consumer does not re-write ONE buffer call after call, and a loop that does measures the memory-side cache
    absorbing the re-writes instead of HBM (one 172.8 MB buffer: 30 us per call; four in turn: 39 us; round 4)."""
    per = batch * 36 * p * n
    k = max(1, min(most, -(-total_bytes // per)))
    return [(torch.empty((batch, 9, p, n), dtype=torch.float32, device=dev),
             torch.empty((batch, p, 3), dtype=torch.int64, device=dev)) for _ in range(k)]


def vox_measure(vox, points, out, pipelined, iters=200, warm=20, kernel_iters=64, ctx=None, dev=None):
    """(seconds per call, kernel means in us, launches timed): wall time with event timing OFF (the event pairs
    cost launch time), the kernels' durations in a pass of their own.  ``out``: one (pillars, indices) pair or a
    list of them used in turn (rotating_outputs).  With a multi-rank ``ctx`` the wall loop is bracketed by barriers
    and the time is the maximum over the ranks (every rank runs the same loop on its own sweeps)."""
    multi = ctx is not None and ctx.distributed
    outs = out if isinstance(out, list) else [out]
    turn = [0]

    def call():
        turn[0] += 1
        o = outs[turn[0] % len(outs)]
        return vox.submit(points, out=o) if pipelined else vox(points, out=o)

    out = outs[0]

    def drain():
        if pipelined:
            for _ in range(vox.LAG):
                vox.submit(None, out=out)
    for _ in range(warm):
        call()
    torch.cuda.synchronize()
    if multi:
        shard.barrier(ctx)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    if multi:
        shard.barrier(ctx)
        torch.cuda.synchronize()
        dt = shard.max_over_ranks(ctx, time.perf_counter() - t1, device=dev) / iters
    else:
        dt = (time.perf_counter() - t1) / iters
    for _ in range(4):
        call()
    torch.cuda.synchronize()
    vox.set_timing(kernel_iters)
    for _ in range(kernel_iters):
        call()
    torch.cuda.synchronize()
    kern_us, n = kernel_means_us(vox)
    vox.set_timing(0)
    drain()
    torch.cuda.synchronize()
    return dt, kern_us, n


def vox_both(vox, points, out, bytes_per_launch, iters=200, kernel_iters=64, ctx=None, dev=None):
    """one voxelizer configuration both ways: software-pipelined (k_step) and as three launches.  With a multi-rank
    ``ctx``: every rank runs it on its own sweeps, `sweeps_per_s` is the WHOLE JOB's (all ranks' sweeps over the
    slowest rank's loop time, `wall_frac` stays per GPU) and `per_rank` carries min / max over the ranks of the
    kernel's duration and fraction."""
    dt_p, k_p, _ = vox_measure(vox, points, out, True, iters=iters, kernel_iters=kernel_iters, ctx=ctx, dev=dev)
    dt_3, k_3, _ = vox_measure(vox, points, out, False, iters=iters, kernel_iters=kernel_iters, ctx=ctx, dev=dev)
    B = points.shape[0]
    W = ctx.world_size if ctx is not None else 1
    rec = {"output_buffers": len(out) if isinstance(out, list) else 1,
           "sweeps_per_s": W * B / dt_p, "us_per_step": dt_p * 1e6,
           "pipeline_GBps": bytes_per_launch / dt_p / 1e9, "wall_frac": bytes_per_launch / dt_p / HBM_PEAK,
           "k_step_us": k_p["k_step"], "kernel_frac": bytes_per_launch / (k_p["k_step"] * 1e-6) / HBM_PEAK,
           "three_launch": dict(three_launch_record(k_3, bytes_per_launch), sweeps_per_s=W * B / dt_3,
                                us_per_step=dt_3 * 1e6, wall_frac=bytes_per_launch / dt_3 / HBM_PEAK)}
    if W > 1:
        rec["n_gpus"] = W
        rec["per_rank"] = ranks_min_max(ctx, k_p["k_step"], bytes_per_launch, dev)
        rec["per_rank"]["kernel_frac"] = rec["per_rank"].pop("frac")
    return rec


def static_traffic(key):
    """HBM bytes per k_emit launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs); a constant read from profiles/, NOT measured in this run"""
    tpath = os.path.join(ROOT, "profiles", "emit_traffic.json")
    try:
        t = json.load(open(tpath))
        return t.get(key), "static: " + t.get("source", "profiles/emit_traffic.json")
    except Exception:
        return None, None


def live_traffic(batch, kernel, timeout_s=90, child="--headline-only"):
    """HBM bytes per launch of `kernel` measured IN THIS RUN: two child runs of this script under `rocprofv3 --pmc`
    -- FETCH_SIZE and WRITE_SIZE in separate passes with --kernel-trace only, from /tmp, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes -- median over the launches of `kernel`, KiB * 1024 (the same
    processing as tools/pmc_summary.py + tools/make_emit_traffic.py).  `child` selects what the child runs: the
    headline loop (`--headline-only`) or the batched target assignment alone (`--targets-only`).  On gfx950
    FETCH_SIZE under-reports 16-byte-per-lane streaming reads by up to 2x (the guide's correction): the caller gets
    the raw figure AND `FETCH_SIZE_bytes_x2` and reports both.  Children are fresh processes in their own process
    group (`-- python3 bench.py ...`, never an exec of this one); on a timeout the whole group is killed and
    reaped, so no profiled child is left holding the GPU.  None when rocprofv3 is missing, fails or times out: the
    caller then falls back to the committed constant and says so."""
    import csv
    import glob
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None                                   # no profiler here, or this run is itself being profiled
    got = {}
    tmp = tempfile.mkdtemp(prefix="pp_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "t", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), child, "--steps", "8", "--warmup", "4",
                   "--batch", str(batch)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)     # the launcher AND the profiled python child
                except ProcessLookupError:
                    pass
                proc.wait()
                return None
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name") or row.get("Kernel-Name") or ""
                    if kernel.replace("pp::", "") in name and row.get("Counter_Name") == ctr:
                        vals.append(float(row["Counter_Value"]))
            if rc != 0 or len(vals) < 4:
                return None
            got[ctr] = (statistics.median(vals) * 1024.0, len(vals))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]
    return total, {"FETCH_SIZE_bytes": got["FETCH_SIZE"][0], "WRITE_SIZE_bytes": got["WRITE_SIZE"][0],
                   "FETCH_SIZE_bytes_x2": 2.0 * got["FETCH_SIZE"][0],
                   "launches": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]]}


def per_rank(ctx, values, device=None):
    """`values` (this rank's floats) from EVERY rank: a list, rank by rank, of lists -- one all-reduce(SUM) of a
    [world, k] tensor in which only this rank's row is non-zero (works over RCCL and over gloo alike; called after a
    timed loop, never inside one)."""
    k = len(values)
    t = torch.zeros((ctx.world_size, k), dtype=torch.float64, device=device)
    t[ctx.rank] = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if ctx.distributed:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return t.cpu().tolist()


def ranks_min_max(ctx, us, bytes_per_launch, device=None, name="k_step_us"):
    """min / max over the ranks of one kernel's mean duration and of its roofline fraction (every rank times its
    own kernel with its own HIP events; north_star asks for the fraction at 1/2/4/8 GPUs)"""
    rows = [r[0] for r in per_rank(ctx, [us], device)]
    fr = [bytes_per_launch / (u * 1e-6) / HBM_PEAK if u > 0 else float("nan") for u in rows]
    return {name: [min(rows), max(rows)], "frac": [min(fr), max(fr)], "per_rank_" + name: rows,
            "what": "this kernel's mean duration on every rank (its own HIP event pairs) and algorithmic bytes / "
                    "duration / 8 TB/s: [min, max] over the ranks"}


def self_launch(n_ranks):
    """Start `n_ranks` rank processes of this script (one per GPU) under torch.distributed.run and
    wait for them.  Called before anything in this process has touched the GPU."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # The ranks inherit this process's environment.  HSA_ENABLE_IPC_MODE_LEGACY=0 is exported by this image and by the
    # GPU boxes (`env | grep HSA_`); where it is ABSENT it is set, never overridden.  Source: the pool's operating notes
    # (the host driver supports dmabuf IPC only; without the variable RCCL / cross-process tensor sharing fails with
    # "hipIpcGetMemHandle: invalid argument") -- taken on trust: RCCL has never run on these one-GPU boxes.
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4,
                    help="sweeps per GPU per step (default: the reference's BATCH_SIZE, config.py:137)")
    ap.add_argument("--mode", choices=["fwd", "train"], default="fwd",
                    help="fwd: BASELINE metric (configs[1]); train: the headline itself becomes "
                         "configs[2] (HIP target assignment, loss forward/backward, all-reduces)")
    ap.add_argument("--three-launch", action="store_true",
                    help="fwd mode: time the headline with the voxelizer as three dependent launches per step "
                         "(pp_voxelize_dev) instead of the software-pipelined k_step")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline loop (no side legs): under rocprofv3 the kernel-stats average of "
                         "k_step then covers exactly the launches `roofline` is computed from")
    ap.add_argument("--targets-only", action="store_true",
                    help="only the batched target assignment of configs[2] (--steps + --warmup launches, no output): "
                         "the child run `train_c3.targets.moved_bytes` is measured from under rocprofv3 --pmc")
    ap.add_argument("--stress-only", action="store_true",
                    help="only configs[4]'s voxelizer loop (k_step on 200k-point sweeps, rotating outputs; --steps + "
                         "--warmup launches, no output): the child run `stress_c5.pipelined.traffic` is measured from")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the dropin_host record (the pybind11 module's "
                                                             "own speed on host arrays)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the next_rows record (lidar ingest, post-processing)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed constant (profiles/emit_traffic.json) instead of two "
                         "rocprofv3 --pmc child runs of the headline loop")
    ap.add_argument("--no-fused", action="store_true", help="skip the fused-feature-net side measurement")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the train_c3 sub-record")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress_c5 sub-record")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the contract); gloo only to rehearse the multi-rank "
                         "code path on a box with fewer GPUs than ranks (all ranks share device 0)")
    a = ap.parse_args()
    if a.headline_only:
        a.no_cpu_baseline = a.no_fused = a.no_train_leg = a.no_stress = a.no_dropin = a.no_next_rows = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has
        # made no HIP call so far and makes none (no torch.cuda.*, no pp_ctx_create): the N ranks are
        # FRESH child processes of torch.distributed.run (never an exec of this one), rank 0's JSON
        # line goes straight through the inherited stdout, and the exit code is the launcher's
        # (non-zero when any rank failed).
        sys.exit(self_launch(a.gpus))

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # a rank started by a launcher other than self_launch: the same default, before this process's first HIP call
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # every rank searches / compiles its convolutions against its own MIOpen user db
        shard.private_miopen_cache(int(os.environ.get("LOCAL_RANK", "0")))
    ctx = shard.init_from_env(a.backend)
    if ctx.world_size != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={ctx.world_size}: launch with "
                         f"torch.distributed.run --nproc-per-node {a.gpus}")
    local_dev = ctx.local_rank if a.backend == "nccl" else ctx.local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # fwd: let MIOpen search its fastest f32 forward algorithms (seconds, in the warm-up).
    # train: immediate mode -- on a fresh box a search over the backward-data / backward-weight
    # solvers first compiles their kernels (5-6 minutes measured, whatever MIOPEN_FIND_MODE says)
    # for a 3 % faster step.
    torch.backends.cudnn.benchmark = (a.mode == "fwd")

    cfg = VoxelConfig.square(HALF, STEP, P, N)          # default pillar order (scrambled)
    if a.targets_only:
        # the PMC child of train_c3.targets: the step's boxes in one launch, nothing else on the device
        from pp_amd import boxes
        from pp_amd.targets import TargetAssigner
        ta_ = TargetAssigner(boxes.AnchorConfig(fm_height=(cfg.canvas_height + 1) // 2,
                                                fm_width=(cfg.canvas_width + 1) // 2),
                             canvas_height=cfg.canvas_height, device=dev)
        g_ = ta_.upload_batch([synth.gt_boxes(40, cfg.canvas_height, s) for s in range(a.batch)])
        o_ = ta_.assign_batch_device(*g_)
        for _ in range(a.warmup + a.steps):
            ta_.assign_batch_device(*g_, out=o_)
        torch.cuda.synchronize()
        shard.shutdown(ctx)
        return
    if a.stress_only:
        # the PMC child of stress_c5: the software-pipelined voxelizer at configs[4]'s shapes, outputs in turn
        c5_ = VoxelConfig.square(C5["half"], C5["step"], C5["P"], C5["N"])
        v5_ = PillarVoxelizer(c5_, device=dev)
        p5_ = torch.from_numpy(np.stack([synth.lidar_like(C5["n"], C5["half"], s) for s in range(a.batch)])).to(dev)
        o5_ = rotating_outputs(a.batch, C5["P"], C5["N"], dev)
        for k_ in range(a.warmup + a.steps + PillarVoxelizer.LAG):
            v5_.submit(p5_, out=o5_[k_ % len(o5_)])
        torch.cuda.synchronize()
        shard.shutdown(ctx)
        return
    pipe = PillarPipeline(cfg, device=dev, seed=0, with_targets=(a.mode == "train"))
    pipe.model.eval() if a.mode == "fwd" else pipe.model.train()
    sweep_ids = [ctx.rank * a.batch + i for i in range(a.batch)]
    clouds = np.stack([synth.lidar_like(N_POINTS, HALF, s) for s in sweep_ids])
    points = torch.from_numpy(clouds).to(dev)          # resident in HBM before timing
    # the pipelined headline hands in a DIFFERENT resident batch every step (four sets in turn: the four batches in
    # flight in k_step are four different clouds, as with a loader); seeds beyond every rank's sweep ids
    point_sets = [points] + [torch.from_numpy(np.stack([synth.lidar_like(N_POINTS, HALF, 1000 * r + s)
                                                         for s in sweep_ids])).to(dev) for r in (1, 2, 3)]
    step_no = [0]
    gts_host = [synth.gt_boxes(40, cfg.canvas_height, s) for s in sweep_ids]
    gts = gts_host
    if a.mode == "train":   # boxes resident on the device like the points (a loader would prefetch them)
        gts = pipe.upload_ground_truth_batch(gts_host)

    def train_step(p_, gts_):
        p_.model.zero_grad(set_to_none=True)
        losses = p_.train_forward_backward(points, gts_, shard_ctx=ctx)
        shard.allreduce_gradients(ctx, p_.model.parameters())     # DataParallel's gradient reduction
        return shard.reduce_loss_scalars(ctx, *losses, n_local=a.batch, device=dev)

    pipelined = a.mode == "fwd" and not a.three_launch

    def step():
        if a.mode == "fwd":
            # software pipeline over consecutive steps (the reference's DataLoader prefetch, train.py:120-121):
            # ONE voxelizer launch = split(batch i) | tile(batch i-1) | order(batch i-2) | emit(batch i-3); the
            # network runs on batch i-3.  Every step does one batch's worth of every stage.
            step_no[0] += 1
            pts_ = point_sets[step_no[0] % len(point_sets)]
            return pipe.forward_pipelined(pts_) if pipelined else pipe.forward(pts_)
        return train_step(pipe, gts)

    def timed_loop(fn, steps):
        torch.cuda.synchronize()
        shard.barrier(ctx)
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        shard.barrier(ctx)
        torch.cuda.synchronize()
        return shard.max_over_ranks(ctx, time.perf_counter() - t_, device=dev)

    # W warm-up steps that reach the network (the first LAG pipelined calls only fill the voxelizer's pipeline)
    for _ in range(a.warmup + (PillarVoxelizer.LAG if pipelined else 0)):
        step()
    pipe.voxelizer.set_timing(min(a.steps, 4096))
    elapsed = timed_loop(step, a.steps)
    kern_us, launches = kernel_means_us(pipe.voxelizer)
    pipe.voxelizer.set_timing(0)
    bytes_per_launch = cfg.algorithmic_bytes(N_POINTS) * a.batch
    head_ranks = ranks_min_max(ctx, kern_us["k_step"] if "k_step" in kern_us else kern_us["k_emit"], bytes_per_launch,
                               dev, name="avg_launch_us")
    three_e2e = None
    if pipelined and a.headline_only:
        for _ in range(PillarVoxelizer.LAG):
            pipe.forward_pipelined(None)                                # drain
    elif pipelined:
        for _ in range(PillarVoxelizer.LAG):
            pipe.forward_pipelined(None)                                # drain
        # the same forward with the voxelizer as three dependent launches per step, for comparison
        def three_step():      # the same four resident batches in turn as the pipelined leg
            step_no[0] += 1
            return pipe.forward(point_sets[step_no[0] % len(point_sets)])
        for _ in range(3):
            three_step()
        pipe.voxelizer.set_timing(min(a.steps, 4096))
        el3 = timed_loop(three_step, a.steps)
        k3, _ = kernel_means_us(pipe.voxelizer)
        pipe.voxelizer.set_timing(0)
        three_e2e = dict(three_launch_record(k3, bytes_per_launch), ms_per_step=el3 / a.steps * 1e3,
                         sweeps_per_s=a.steps * a.batch * ctx.world_size / el3)

    # The same forward with the voxelizer's launch on a second stream beside the network (forward_overlapped:
    # the reference's DataLoader workers run beside the model, train.py:120-121).  Reported with the step
    # time AND the kernel's duration: an overlapped kernel shares the chip with MFMA-bound convolutions and
    # reads longer, while the step loses the voxelizer from its critical path.
    overlap = None
    if pipelined and not a.headline_only:
        def ov_step():
            step_no[0] += 1
            return pipe.forward_overlapped(point_sets[step_no[0] % len(point_sets)])
        for _ in range(3 + PillarVoxelizer.LAG + 1):
            ov_step()
        pipe.voxelizer.set_timing(min(a.steps, 4096))
        el_o = timed_loop(ov_step, a.steps)
        ko, _ = kernel_means_us(pipe.voxelizer)
        pipe.voxelizer.set_timing(0)
        for _ in range(PillarVoxelizer.LAG + 1):
            pipe.forward_overlapped(None)
        torch.cuda.synchronize()
        overlap = {"value": a.steps * a.batch * ctx.world_size / el_o, "unit": "sweeps/s",
                   "ms_per_step": el_o / a.steps * 1e3, "latency_calls": PillarVoxelizer.LAG + 1,
                   "k_step_us_while_overlapped": ko.get("k_step"),
                   "k_step_frac_while_overlapped": bytes_per_launch / (ko["k_step"] * 1e-6) / HBM_PEAK,
                   "what": "PillarPipeline.forward_overlapped: k_step on a side stream, two output buffers, the network "
                           "on the caller's stream consumes the batch the previous call's launch emitted"}

    # voxelizer alone (same resident inputs), for the per-stage picture: wall time per call (event timing off)
    # and kernel durations, software-pipelined and as three launches; default order and the row-major one; ONE
    # sweep per launch (BASELINE configs[3]'s per-GPU shape); BASELINE configs[0]'s 100x100 grid
    vox_rec = None
    if not a.headline_only:
        rot_b = rotating_outputs(a.batch, P, N, dev)
        vox_rec = vox_both(pipe.voxelizer, points, rot_b, bytes_per_launch)
        vox_rec["what"] = ("wall time per voxelizer call, calls back to back, outputs into " + str(len(rot_b)) +
                           " buffers in turn (>= 512 MB together: no buffer is still in the 256 MB Infinity Cache when it "
                           "is written again -- every byte goes to HBM, as in the end-to-end loop)")
        sb_ = vox_both(pipe.voxelizer, points, pipe._buffers(a.batch), bytes_per_launch, iters=100)
        vox_rec["same_output_buffer"] = dict(
            {k_: sb_[k_] for k_ in ("sweeps_per_s", "us_per_step", "wall_frac", "k_step_us", "kernel_frac")},
            three_launch_us_per_step=sb_["three_launch"]["us_per_step"],
            what="rounds 1-3's definition: ONE output buffer re-written call after call -- the memory-side cache absorbs "
                 "the re-writes of a buffer smaller than itself, so this is not an HBM figure; kept for comparison")
        vox_rm = PillarVoxelizer(VoxelConfig.square(HALF, STEP, P, N, order=_lib.ORDER_ROW_MAJOR), device=dev)
        vox_rec["row_major_order"] = vox_both(vox_rm, points, rot_b, bytes_per_launch, iters=100)
        del vox_rm
        vox_rec["one_sweep_per_launch"] = vox_both(pipe.voxelizer, points[:1], rotating_outputs(1, P, N, dev),
                                                   bytes_per_launch // a.batch)
        c1cfg = VoxelConfig.square(C1["half"], C1["step"], C1["P"], C1["N"])
        vox_c1 = PillarVoxelizer(c1cfg, device=dev)
        vox_rec["c1_shapes"] = dict(
            vox_both(vox_c1, points, rot_b, c1cfg.algorithmic_bytes(C1["n"]) * a.batch, iters=100),
            what="BASELINE configs[0]'s shapes on the GPU: the same clouds on the 100x100 grid (1 m cells, up to "
                 "~380 points in a cell: long sequential running-mean chains, pillars.cpp:311-328)")
        del vox_c1, rot_b
        torch.cuda.empty_cache()

    # next row: the feature net fused into the voxelizer -- the dense
    # [9,P,N] tensor is never built.  Reported beside the headline, not as it.
    fused = None
    if a.mode == "fwd" and not a.no_fused:
        for _ in range(max(3, a.warmup // 2)):
            pipe.forward_fused(points)
        torch.cuda.synchronize()
        shard.barrier(ctx)
        t2 = time.perf_counter()
        for _ in range(a.steps):
            pipe.forward_fused(points)
        torch.cuda.synchronize()
        shard.barrier(ctx)
        f_el = shard.max_over_ranks(ctx, time.perf_counter() - t2, device=dev)
        fused = {"value": a.steps * a.batch * ctx.world_size / f_el, "unit": "sweeps/s",
                 "ms_per_step": f_el / a.steps * 1e3,
                 "what": "same forward with PPFeatureNet (conv1x1+ReLU+BN(eval)+max) and PPScatter fused into "
                         "the HIP voxelizer (pp_voxelize_pfn_canvas_dev, channels-last canvas); outputs equal "
                         "the headline path's within 1e-4"}
        # ... and the fused voxelizer call by itself against ITS bytes: points in, one 256-byte feature pixel
        # per pillar, the indices, and the canvas clear (the dense [9,P,N] tensor does not exist here)
        pfn_tab = pipe.model.feature_net.fused_table(dev)
        H_, W_ = pipe.model.scatter.h, pipe.model.scatter.w
        f_out = (pipe._canvas(a.batch, H_, W_), torch.empty((a.batch, P, 3), dtype=torch.int64, device=dev))
        pipe._canvas_key = None      # the pipeline's record of the canvas's non-zero pixels ends here

        def fused_wall(reuse):
            for _ in range(10):
                pipe.voxelizer.pfn_canvas(points, pfn_tab, (H_, W_), out=f_out, reuse=reuse)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(100):
                pipe.voxelizer.pfn_canvas(points, pfn_tab, (H_, W_), out=f_out, reuse=reuse)
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / 100
        f_dt_full = fused_wall(False)            # unknown canvas: all 64 MB per sweep cleared
        f_dt = fused_wall(True)                  # the canvas handed back: only the previous call's pixels cleared
        pipe.voxelizer.set_timing(64)
        for _ in range(64):
            pipe.voxelizer.pfn_canvas(points, pfn_tab, (H_, W_), out=f_out, reuse=True)
        torch.cuda.synchronize()
        f_k, _ = kernel_means_us(pipe.voxelizer)
        pipe.voxelizer.set_timing(0)
        f_live = 16 * N_POINTS + 256 * P + 24 * P                      # per sweep: points in, pixels + indices out
        f_sparse = 256 * P + 24 * P                                    # ... the previous call's pixels zeroed, its indices read
        f_clear = 64 * H_ * W_ * 4                                     # ... or the whole canvas
        fused["roofline"] = {
            "bound": "hbm by its bytes; measured: latency-bound (three dependent launches, 4 waves per SIMD)",
            "kernel": "pp::k_emit<float,3,0> (+ k_split, k_tile with the sparse canvas clear in extra workgroups)",
            "bytes_per_launch": (f_live + f_sparse) * a.batch,
            "bytes_what": "per sweep 16*n + 256*P (one channels-last feature pixel per pillar) + 24*P (indices) + "
                          "256*P + 24*P (the previous call's pixels zeroed: pp_voxelize_pfn_canvas_reuse_dev)",
            "us_per_call": f_dt * 1e6, "kernels_us": f_k, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "achieved": (f_live + f_sparse) * a.batch / f_dt / 1e9,
            "frac": (f_live + f_sparse) * a.batch / f_dt / HBM_PEAK,
            "k_emit_frac_of_its_own_bytes": f_live * a.batch / (f_k["k_emit"] * 1e-6) / HBM_PEAK,
            "full_clear": {"us_per_call": f_dt_full * 1e6, "bytes_per_launch": (f_live + f_clear) * a.batch,
                           "frac": (f_live + f_clear) * a.batch / f_dt_full / HBM_PEAK,
                           "what": "pp_voxelize_pfn_canvas_dev: an unknown canvas, all 64*H*W*4 bytes per sweep "
                                   "cleared by a memset first"}}

    if fused is not None:
        # ... and the fused path in its one-launch form (k_step<pfn>: split | tile | order | fused emit | clear)
        def fp_step():
            step_no[0] += 1
            return pipe.forward_fused_pipelined(point_sets[step_no[0] % len(point_sets)])
        for _ in range(3 + PillarVoxelizer.LAG):
            fp_step()
        el_f = timed_loop(fp_step, a.steps)
        for _ in range(PillarVoxelizer.LAG):
            pipe.forward_fused_pipelined(None)
        for _ in range(10):
            pipe.voxelizer.submit_pfn_canvas(points, pfn_tab, (H_, W_))
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(200):
            pipe.voxelizer.submit_pfn_canvas(points, pfn_tab, (H_, W_))
        torch.cuda.synchronize()
        fp_dt = (time.perf_counter() - t_) / 200
        pipe.voxelizer.set_timing(64)
        for _ in range(64):
            pipe.voxelizer.submit_pfn_canvas(points, pfn_tab, (H_, W_))
        torch.cuda.synchronize()
        fp_k, _ = kernel_means_us(pipe.voxelizer)
        pipe.voxelizer.set_timing(0)
        pipe.voxelizer.reset_stream()
        fused["pipelined"] = {
            "value": a.steps * a.batch * ctx.world_size / el_f, "unit": "sweeps/s", "ms_per_step": el_f / a.steps * 1e3,
            "roofline": {"kernel": "pp::k_step<3, 0> (one launch per call: split | tile | order roles, the emit role as "
                                   "the fused feature net + scatter, a clear role for the other of two canvases)",
                         "bytes_per_launch": (f_live + f_sparse) * a.batch, "us_per_call": fp_dt * 1e6,
                         "k_step_us": fp_k.get("k_step"), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "achieved": (f_live + f_sparse) * a.batch / fp_dt / 1e9,
                         "frac": (f_live + f_sparse) * a.batch / fp_dt / HBM_PEAK},
            "what": "PillarPipeline.forward_fused_pipelined / pp_voxelize_step_pfn_canvas_dev; outputs equal the "
                    "three-launch fused call's bit for bit, LAG calls later"}

    # BASELINE configs[2] (and configs[3]'s collectives when N > 1): target assignment + loss
    # forward/backward + the gradient and loss-scalar all-reduces, every rank, timed like the headline
    train = None
    if a.mode == "fwd" and not a.no_train_leg:
        torch.backends.cudnn.benchmark = False          # immediate mode, see above
        tp = PillarPipeline(cfg, device=dev, seed=0, with_targets=True)
        tp.model.train()
        tg = tp.upload_ground_truth_batch(gts_host)      # the step's boxes: one buffer, one launch for the batch
        for _ in range(3):
            train_step(tp, tg)
        torch.cuda.synchronize()
        shard.barrier(ctx)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(a.train_steps):
            train_step(tp, tg)
        torch.cuda.synchronize()
        shard.barrier(ctx)
        torch.cuda.synchronize()
        tr_el = shard.max_over_ranks(ctx, time.perf_counter() - t3, device=dev)
        # the pieces, on this rank: target assignment of the batch, the two all-reduces
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_out = tp.assigner.assign_batch_device(*tg)
        for _ in range(10):
            tp.assigner.assign_batch_device(*tg, out=t_out)
        e0.record()
        for _ in range(100):
            tp.assigner.assign_batch_device(*tg, out=t_out)
        e1.record()
        torch.cuda.synchronize()
        assign_call_us = e0.elapsed_time(e1) * 1e3 / 100
        assign_us = assign_call_us / a.batch
        tg1 = [tp.upload_ground_truth(g) for g in gts_host]     # ... and one launch per sample (rounds 1-3)
        # (outputs re-used: with two allocator calls per sample the loop is bound by the HOST's issue rate on this pool's
        # slower boxes -- 12.3 us per call issued against an 11.5 us kernel where measured, 16-19 us read on some boxes in
        # round 5 while the kernel trace showed 13; profiles/r06/NOTES.md)
        o1 = (torch.empty((tp.assigner.A, tp.assigner.num_classes), dtype=torch.float32, device=dev),
              torch.empty((tp.assigner.A, 9), dtype=torch.float32, device=dev))
        for g in tg1:
            tp.assigner.assign_device(*g, out=o1)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        e0.record()
        for _ in range(20):
            for g in tg1:
                tp.assigner.assign_device(*g, out=o1)
        e1.record()
        assign1_issue_us = (time.perf_counter() - t_h) * 1e6 / (20 * len(tg1))
        torch.cuda.synchronize()
        assign1_us = e0.elapsed_time(e1) * 1e3 / (20 * len(tg1))
        t_bytes = 112 * tp.assigner.A * a.batch
        targets_rec = {"bound": "hbm by its bytes; measured: a chain of latencies (box loads, gate, clip rounds of f64, rows, "
                                "the last-workgroup tail) beside a zero fill -- see DESIGN.md",
                       "kernel": "pp::k_targets_gt (box-centric: grid = samples x {zero-fill workgroups, workgroups per box}; one "
                                 "launch per step; anchor ARRAYS, and more than 8 anchor types per cell, go through the "
                                 "anchor-centric pp::k_targets<false>)",
                       "bytes_per_launch": t_bytes, "bytes_what": "112 * A per sample (SURVEY 8d) x the batch",
                       "us_per_call": assign_call_us, "achieved": t_bytes / (assign_call_us * 1e-6) / 1e9,
                       "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": t_bytes / (assign_call_us * 1e-6) / HBM_PEAK,
                       "one_sample_per_launch_us": assign1_us, "one_sample_host_issue_us": assign1_issue_us,
                       "one_sample_per_launch_frac": 112 * tp.assigner.A / (assign1_us * 1e-6) / HBM_PEAK,
                       # what the box-centric kernel MOVES: it evaluates the anchors arithmetically and reads none of
                       # SURVEY's 40 B per anchor of input -- the two target arrays (72 B per anchor) and the boxes
                       "moved_bytes": 72 * tp.assigner.A * a.batch,
                       "moved_bytes_source": "computed: 72 * A * batch (the two f32 target arrays; the boxes stay in L2)",
                       "frac_of_moved_bytes": 72 * tp.assigner.A * a.batch / (assign_call_us * 1e-6) / HBM_PEAK,
                       "one_sample_frac_of_moved_bytes": 72 * tp.assigner.A / (assign1_us * 1e-6) / HBM_PEAK}
        if ctx.world_size == 1 and not a.no_live_traffic:
            # ... and what the counters say it moves, measured in this run (two rocprofv3 --pmc child runs of the batched
            # assignment alone; None where the profiler is missing or this run is itself being profiled)
            torch.cuda.synchronize()
            lt = live_traffic(a.batch, "k_targets_gt", child="--targets-only")
            if lt is not None:
                targets_rec["traffic"] = lt[0]
                targets_rec["traffic_detail"] = lt[1]
                targets_rec["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                                 "passes over `bench.py --targets-only`, median per launch, KiB * 1024")
                targets_rec["frac_of_measured_bytes"] = lt[0] / (assign_call_us * 1e-6) / HBM_PEAK
            else:
                targets_rec["traffic"] = None
        del tg1, t_out
        e0.record()
        for _ in range(10):
            shard.allreduce_gradients(ctx, tp.model.parameters())
            shard.reduce_loss_scalars(ctx, 0.0, 0.0, 0.0, 0.0, n_local=a.batch, device=dev)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 10
        nbytes = sum(p_.numel() * p_.element_size() for p_ in tp.model.parameters() if p_.requires_grad)
        train = {"value": a.train_steps * a.batch * ctx.world_size / tr_el, "unit": "sweeps/s",
                 "ms_per_step": tr_el / a.train_steps * 1e3, "steps": a.train_steps,
                 "target_assign_us_per_sweep": assign_us, "targets": targets_rec,
                 "allreduce_ms": ar_ms if ctx.distributed else 0.0,
                 "collectives": {"backend": ctx.backend if ctx.distributed else None,
                                 "world_size": torch.distributed.get_world_size() if ctx.distributed else 1,
                                 "per_step": "1 all-reduce of the positive-anchor counts (4 B), 1 flat gradient "
                                             f"all-reduce ({nbytes / 1e6:.1f} MB f32), 1 all-reduce of the four "
                                             "loss scalars + sweep count (20 B)" if ctx.distributed else "none (N=1)"},
                 "what": "configs[2]: HIP voxelizer + HIP rotated-IoU target assignment (2 anchors/cell, "
                         "G=40 boxes, A=125000) + network forward + focal/smooth-L1 loss + backward, f32, "
                         "BatchNorm in training mode; MIOpen immediate mode"}
        if ctx.world_size > 1:
            # BASELINE configs[3] as named: batch = N sweeps sharded ONE per GPU (global batch = world size), the
            # forward and the training step with its collectives, so that a SCALE run measures that config too
            one, g_one = points[:1], tp.upload_ground_truth_batch(gts_host[:1])
            for _ in range(3 + PillarVoxelizer.LAG):
                pipe.forward_pipelined(one)
            pipe.voxelizer.set_timing(min(a.steps, 4096))
            el1 = timed_loop(lambda: pipe.forward_pipelined(one), a.steps)
            k1, n1 = kernel_means_us(pipe.voxelizer)
            pipe.voxelizer.set_timing(0)
            for _ in range(PillarVoxelizer.LAG):
                pipe.forward_pipelined(None)
            b1 = bytes_per_launch // a.batch
            c3_roof = dict(roofline_record(k1, n1, b1, step_kernel=pipe.voxelizer.step_kernel_name(1)),
                           ranks_min_max=ranks_min_max(ctx, k1["k_step"], b1, dev, name="avg_launch_us"))
            c3_roof["what"] = ("k_step at ONE sweep per launch inside the one-sweep-per-GPU forward loop (HIP event pairs, "
                               "every rank its own)")
            to1 = tp.assigner.assign_batch_device(*g_one)
            for _ in range(10):
                tp.assigner.assign_batch_device(*g_one, out=to1)
            e0.record()
            for _ in range(100):
                tp.assigner.assign_batch_device(*g_one, out=to1)
            e1.record()
            torch.cuda.synchronize()
            tg1_us = e0.elapsed_time(e1) * 1e3 / 100
            c3_tgt = {"kernel": "pp::k_targets_gt, one sample per launch", "us_per_call": tg1_us,
                      "bytes_per_launch": 112 * tp.assigner.A, "moved_bytes": 72 * tp.assigner.A,
                      "frac": 112 * tp.assigner.A / (tg1_us * 1e-6) / HBM_PEAK,
                      "frac_of_moved_bytes": 72 * tp.assigner.A / (tg1_us * 1e-6) / HBM_PEAK,
                      "ranks_min_max": ranks_min_max(ctx, tg1_us, 112 * tp.assigner.A, dev, name="us_per_call")}
            del to1

            def train_one():
                tp.model.zero_grad(set_to_none=True)
                losses = tp.train_forward_backward(one, g_one, shard_ctx=ctx)
                shard.allreduce_gradients(ctx, tp.model.parameters())
                return shard.reduce_loss_scalars(ctx, *losses, n_local=1, device=dev)
            for _ in range(2):
                train_one()
            elt = timed_loop(train_one, a.train_steps)
            train["configs3_one_sweep_per_gpu"] = {
                "global_batch": ctx.world_size, "sweeps_per_gpu_per_step": 1,
                "forward": {"value": a.steps * ctx.world_size / el1, "unit": "sweeps/s",
                            "ms_per_step": el1 / a.steps * 1e3, "steps": a.steps},
                "train": {"value": a.train_steps * ctx.world_size / elt, "unit": "sweeps/s",
                          "ms_per_step": elt / a.train_steps * 1e3, "steps": a.train_steps},
                "allreduce_ms": ar_ms,
                "roofline": c3_roof, "targets": c3_tgt,
                "collectives": {"backend": ctx.backend, "world_size": torch.distributed.get_world_size(),
                                "in_forward": "none (sweeps shard; no data-path collective)",
                                "in_train_step": "positive-count all-reduce (4 B), flat gradient all-reduce "
                                                 f"({nbytes / 1e6:.1f} MB f32), loss-scalar all-reduce (20 B)"},
                "what": "configs[3]: one sweep per GPU and step; whole-job sweeps/s over all ranks (max over ranks of "
                        "the loop time)"}
            del one, g_one
        del tp, tg
        torch.cuda.empty_cache()

    # BASELINE configs[4] shapes (stress): voxelizer only, one GPU's share -- B sweeps per launch and the
    # configs[4] per-GPU shape (ONE 200k-point sweep per launch), both ways
    stress = None
    if not a.no_stress:
        # every rank: its own sweeps (seed = global sweep id), its own rotating outputs; at N > 1 the wall loops are
        # bracketed by barriers (sweeps_per_s = all ranks' sweeps over the slowest rank's time) and `per_rank` carries
        # min / max over the ranks of the kernel's duration and fraction -- no collective inside a loop
        c5 = VoxelConfig.square(C5["half"], C5["step"], C5["P"], C5["N"])
        v5 = PillarVoxelizer(c5, device=dev)
        pts5 = torch.from_numpy(np.stack([synth.lidar_like(C5["n"], C5["half"], s) for s in sweep_ids])).to(dev)
        out5 = rotating_outputs(a.batch, C5["P"], C5["N"], dev)
        b5 = c5.algorithmic_bytes(C5["n"]) * a.batch
        r5 = vox_both(v5, pts5, out5, b5, iters=100, ctx=ctx, dev=dev)
        dt3, k3 = r5["three_launch"]["us_per_step"] * 1e-6, r5["three_launch"]["kernels_us"]
        best = min(dt3, r5["us_per_step"] * 1e-6)
        on = "one GPU" if ctx.world_size == 1 else f"each of {ctx.world_size} GPUs"
        stress = {"workload": f"configs[4] shapes on {on}: {a.batch} x {C5['n']}-pt clouds, 1000x1000 grid, "
                              f"P={C5['P']} N={C5['N']}, voxelizer only",
                  # beyond the Infinity Cache the three-launch path can be the faster one at this batch (the
                  # binning stages are issue-bound at 800k points and k_step's roles share the workgroup slots)
                  "sweeps_per_s": ctx.world_size * a.batch / best, "us_per_step": best * 1e6,
                  "wall_frac": b5 / best / HBM_PEAK,
                  "roofline": roofline_record(k3, 64, b5, *static_traffic(f"c5_batch{a.batch}")),
                  "pipelined": {k: r5[k] for k in ("sweeps_per_s", "us_per_step", "wall_frac", "k_step_us",
                                                   "kernel_frac")},
                  "three_launch_us_per_step": dt3 * 1e6,
                  "output_buffers": len(out5),
                  "one_sweep_per_launch": vox_both(v5, pts5[:1], rotating_outputs(1, C5["P"], C5["N"], dev),
                                                   b5 // a.batch, iters=100, ctx=ctx, dev=dev)}
        stress["roofline"]["traffic_what"] = "the three-launch path's k_emit: " + str(stress["roofline"].get("traffic_source"))
        if ctx.world_size == 1 and not a.no_live_traffic:
            # HBM bytes per k_step launch at these shapes, measured in this run like the headline's (two rocprofv3 --pmc
            # child runs of `bench.py --stress-only`)
            torch.cuda.synchronize()
            lt5 = live_traffic(a.batch, v5.step_kernel_name(a.batch), child="--stress-only", timeout_s=120)
            if lt5 is not None:
                stress["pipelined"].update(
                    traffic=lt5[0], traffic_detail=lt5[1], traffic_over_algorithmic=lt5[0] / b5,
                    traffic_fetch_x2=lt5[1]["FETCH_SIZE_bytes_x2"] + lt5[1]["WRITE_SIZE_bytes"],
                    traffic_source="measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
                                   "`bench.py --stress-only`, median per k_step launch, KiB * 1024")
            else:
                stress["pipelined"]["traffic"] = None
        if ctx.world_size > 1:
            stress["n_gpus"] = ctx.world_size
            stress["per_rank"] = r5["per_rank"]
        del v5, pts5, out5
        torch.cuda.empty_cache()
        # ... and configs[4]'s canvas END TO END (SURVEY section 7 hard part 8: up3's output_padding is 3 at 1000x1000):
        # the pipelined forward, voxelizer + network, on the 200k-point sweeps
        if a.mode == "fwd":
            torch.backends.cudnn.benchmark = True
            p5 = PillarPipeline(c5, device=dev, seed=0)
            p5.model.eval()
            b_e = min(a.batch, 2)
            pts_e = [torch.from_numpy(np.stack([synth.lidar_like(C5["n"], C5["half"], 2000 * r + s)
                                                for s in sweep_ids[:b_e]])).to(dev) for r in range(2)]
            turn5 = [0]

            def step5():
                turn5[0] += 1
                return p5.forward_pipelined(pts_e[turn5[0] % 2])
            for _ in range(3 + PillarVoxelizer.LAG):
                step5()
            n5 = max(4, a.steps // 2)
            p5.voxelizer.set_timing(n5)
            el5 = timed_loop(step5, n5)
            k5, _ = kernel_means_us(p5.voxelizer)
            p5.voxelizer.set_timing(0)
            for _ in range(PillarVoxelizer.LAG):
                p5.forward_pipelined(None)
            b5e = c5.algorithmic_bytes(C5["n"]) * b_e
            stress["end_to_end"] = {
                "value": n5 * b_e * ctx.world_size / el5, "unit": "sweeps/s", "ms_per_step": el5 / n5 * 1e3, "steps": n5,
                "sweeps_per_gpu_per_step": b_e, "k_step_us": k5.get("k_step"),
                "k_step_frac": b5e / (k5["k_step"] * 1e-6) / HBM_PEAK,
                "up3_output_padding": int(p5.model.backbone.up3.conv2d_t.output_padding[0]),
                "what": "configs[4]'s sizes end to end: HIP voxelizer (k_step) + PPFeatureNet + scatter + backbone + head "
                        "forward at the 1000x1000 canvas, f32, inference, software-pipelined like the headline"}
            del p5, pts_e
            torch.cuda.empty_cache()

    # the reference's SHIPPED configuration (config.py:46-61,109-123): 600x600 grid, P=24000, N=200 (every
    # bucket cap above the 128-point LDS pool, 172.8 MB per sweep), 6 anchors per cell = 540 000 anchors
    refdef = None
    if not a.no_stress and ctx.world_size == 1:
        from pp_amd import boxes
        from pp_amd.targets import TargetAssigner
        rc_ = VoxelConfig.reference_default()
        vr = PillarVoxelizer(rc_, device=dev)
        ptsr = torch.from_numpy(np.stack([synth.lidar_like(N_POINTS, 60.0, s) for s in sweep_ids])).to(dev)
        outr = rotating_outputs(a.batch, rc_.max_pillars, rc_.max_points_per_pillar, dev)
        br = rc_.algorithmic_bytes(N_POINTS) * a.batch
        refdef = {"workload": f"config.py defaults: {a.batch} x {N_POINTS}-pt clouds, 600x600 grid, P=24000 N=200, "
                              "voxelizer only; target assignment at 300x300x6 = 540000 anchors, G=40",
                  "voxelizer": vox_both(vr, ptsr, outr, br, iters=60),
                  "one_sweep_per_launch": vox_both(vr, ptsr[:1],
                                                   rotating_outputs(1, rc_.max_pillars, rc_.max_points_per_pillar, dev),
                                                   br // a.batch, iters=100)}
        del vr, ptsr, outr
        torch.cuda.empty_cache()
        ta = TargetAssigner(boxes.AnchorConfig.reference_default(), canvas_height=600, device=dev)
        gr = synth.gt_boxes(40, 600, 0)
        g_ = ta._gt_to_device(gr["centers"], gr["wlh"], gr["yaw"], gr["classes"])
        or_ = (torch.empty((ta.A, ta.num_classes), dtype=torch.float32, device=dev),
               torch.empty((ta.A, 9), dtype=torch.float32, device=dev))
        for _ in range(10):
            ta.assign_device(*g_, out=or_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ta.assign_device(*g_, out=or_)
        e1.record()
        torch.cuda.synchronize()
        t_us = e0.elapsed_time(e1) * 1e3 / 100
        refdef["target_assign_us"] = t_us
        refdef["target_assign_frac"] = 112 * ta.A / (t_us * 1e-6) / HBM_PEAK
        gb_ = ta.upload_batch([synth.gt_boxes(40, 600, s) for s in range(a.batch)])   # the step's batch in one launch
        ob_ = ta.assign_batch_device(*gb_)
        for _ in range(10):
            ta.assign_batch_device(*gb_, out=ob_)
        e0.record()
        for _ in range(50):
            ta.assign_batch_device(*gb_, out=ob_)
        e1.record()
        torch.cuda.synchronize()
        tb_us = e0.elapsed_time(e1) * 1e3 / 50
        refdef["target_assign_batch"] = {"samples": a.batch, "us_per_call": tb_us, "us_per_sample": tb_us / a.batch,
                                         "frac": 112 * ta.A * a.batch / (tb_us * 1e-6) / HBM_PEAK}
        del gb_, ob_
        del ta
        torch.cuda.empty_cache()

    if ctx.rank == 0:
        total_sweeps = a.steps * a.batch * ctx.world_size
        traffic, tsrc = static_traffic(f"step_batch{a.batch}" if pipelined else f"batch{a.batch}")
        traffic_live = None
        if pipelined and ctx.world_size == 1 and a.mode == "fwd" and not a.headline_only and not a.no_live_traffic:
            torch.cuda.synchronize()
            traffic_live = live_traffic(a.batch, pipe.voxelizer.step_kernel_name(a.batch))
        out = {
            "metric": METRIC, "value": total_sweeps / elapsed, "unit": "sweeps/s",
            "n_gpus": ctx.world_size, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: HIP pillarizer + PPFeatureNet + scatter + backbone + "
                                   "head fwd, 60k-pt lidar-like clouds, 500x500 grid, P=12000 N=100 D=9, "
                                   "random weights" + ("; configs[2] additions: HIP target assign "
                                                       "(2 anchors/cell, G=40), loss fwd/bwd, gradient and "
                                                       "loss-scalar all-reduces" if a.mode == "train" else ""),
                       "mode": a.mode, "sweeps_per_gpu_per_step": a.batch,
                       "global_batch": a.batch * ctx.world_size,
                       "pillar_order": "scrambled (default; stand-in for the reference's hash-map order)",
                       "voxelizer_arithmetic": "f64 binning/mean, f32 features",
                       "parallelism": f"1 sweep-shard per GPU x{ctx.world_size}, no data-path collective"},
            "collectives": {"backend": ctx.backend if ctx.distributed else None,
                            "world_size": torch.distributed.get_world_size() if ctx.distributed else 1,
                            "in_timed_step": "none (sweeps shard; no data-path collective)" if a.mode == "fwd"
                            else "positive-count, gradient and loss-scalar all-reduces"},
            "roofline": dict(roofline_record(kern_us, launches, bytes_per_launch, traffic, tsrc, three=three_e2e,
                                             step_kernel=pipe.voxelizer.step_kernel_name(a.batch)),
                             ranks_min_max=head_ranks),
            # the headline's definition, for comparisons across rounds: a software pipeline whose outputs lag
            # `latency_calls` calls; `three_launch_value` is the same forward with three dependent launches per step
            "pipelined": bool(pipelined), "latency_calls": PillarVoxelizer.LAG if pipelined else 0,
            "three_launch_value": three_e2e["sweeps_per_s"] if three_e2e else None,
            "distinct_resident_batches": len(point_sets) if a.mode == "fwd" else 1,
        }
        if traffic_live is not None:
            out["roofline"]["traffic_static"] = traffic
            out["roofline"]["traffic"] = traffic_live[0]
            out["roofline"]["traffic_detail"] = traffic_live[1]
            # MI355X_MICROARCH.md's gfx950 correction: FETCH_SIZE under-reports 16-byte-per-lane streaming reads (the
            # split and prefetch roles' float4 streams) by up to 2x -- the reading with FETCH doubled, beside the raw one
            out["roofline"]["traffic_fetch_x2"] = traffic_live[1]["FETCH_SIZE_bytes_x2"] + traffic_live[1]["WRITE_SIZE_bytes"]
            out["roofline"]["traffic_source"] = (
                "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child runs of "
                "`bench.py --headline-only`, --kernel-trace only), median per launch of the headline's k_step "
                "instance, KiB * 1024; `traffic_static` = the committed constant of profiles/emit_traffic.json")
        if overlap is not None:
            out["overlapped"] = overlap
        if vox_rec is not None:
            out["voxelizer_only"] = vox_rec
        out["config"]["voxelizer"] = (
            "software-pipelined over consecutive steps: ONE launch per step (k_step) runs the split stage of "
            "batch i, the tile stage of batch i-1, the order stage of batch i-2 and the emit stage of batch i-3 side "
            "by side; the network consumes batch i-3 (PillarPipeline.forward_pipelined).  Every step does one batch's worth of every "
            "stage; `roofline.three_launch` is the same forward with three dependent launches per step"
            if pipelined else "three dependent launches per step (pp_voxelize_dev: k_split, k_tile, k_emit)")
        if fused is not None:
            out["fused_feature_net"] = fused
        if train is not None:
            out["train_c3"] = train
        if stress is not None:
            out["stress_c5"] = stress
        if refdef is not None:
            out["reference_default"] = refdef
        if not a.no_next_rows and ctx.world_size == 1 and a.mode == "fwd":
            out["next_rows"] = next_rows_record(pipe, points, dev)
        if not a.no_dropin and not a.headline_only and ctx.world_size == 1:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["dropin_host"] = dropin_host()
        if not a.no_cpu_baseline and ctx.world_size == 1:
            out["cpu_baseline"] = cpu_baseline()
        elif ctx.world_size > 1:
            # the CPU legs are timed by rank 0 at N = 1 only (while N rank processes hold the host's cores the figure would
            # be a different one): the line to read them from
            out["cpu_baseline_from"] = "the N = 1 line of the same run series (`python bench.py`): cpu_baseline.port / .faithful"
        print(json.dumps(out))
    shard.barrier(ctx)
    shard.shutdown(ctx)


if __name__ == "__main__":
    main()
