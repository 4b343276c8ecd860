"""profiles/emit_traffic.json from the committed PMC summaries of the current round (profiles/<round>/pmc_*_summary.csv):
HBM bytes per launch of the dominant voxelizer kernel = FETCH_SIZE + WRITE_SIZE medians (KiB * 1024).
bench.py prints these constants as `roofline.traffic` (labelled static: they are not measured in the bench run)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
D = os.path.join(ROOT, "profiles", rnd)


def read(name):
    out = {}
    path = os.path.join(D, name + "_summary.csv")
    if not os.path.exists(path):
        return out
    for r in csv.reader(open(path)):
        if r and r[0] != "kernel":
            out[(r[0].replace("void ", ""), r[1])] = (int(r[2]), float(r[3]) * 1024.0)
    return out


def kernel_bytes(tag, prefix):
    f, w = read(f"pmc_{tag}_fetch"), read(f"pmc_{tag}_write")
    det = {}
    for (k, _), (n, v) in f.items():
        if n >= 10:
            det.setdefault(k, {})["FETCH_SIZE_bytes"] = v
    for (k, _), (n, v) in w.items():
        if n >= 10:
            det.setdefault(k, {})["WRITE_SIZE_bytes"] = v
    main = [k for k in det if k.startswith(prefix)]
    if not main:
        return None, det
    m = det[main[0]]
    return m.get("FETCH_SIZE_bytes", 0.0) + m.get("WRITE_SIZE_bytes", 0.0), det


out = {"source": f"profiles/{rnd}/pmc_*_summary.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                 "tools/bench_vox.py (tools/collect_profiles.sh), this round's kernels; a constant read by bench.py, "
                 "not measured in the bench run",
       "note": "HBM bytes per launch = FETCH_SIZE + WRITE_SIZE medians (KiB * 1024) of the dominant kernel: k_step "
               "(software-pipelined calls: the whole voxelizer incl. its prefetch role) or k_emit (three-launch calls); "
               "their reads are narrow gathers / short rows, so the guide's x2 correction for wide streaming reads "
               "is not applied; step_batch4 comes from the PMC passes over bench.py's own headline loop "
               "(`bench.py --headline-only`), step_batch4_loop from the voxelizer-only loop"}
for key, tag, prefix in (("step_batch4_loop", "c2_b4_step", "pp::k_step"), ("step_batch4", "bench_headline", "pp::k_step"), ("step_batch1", "c2_b1_step", "pp::k_step"),
                         ("batch4", "c2_b4_three", "pp::k_emit"), ("batch1", "c2_b1_three", "pp::k_emit"),
                         ("c5_batch4", "c5_b4_three", "pp::k_emit"), ("c5_step_batch1", "c5_b1_step", "pp::k_step")):
    tot, det = kernel_bytes(tag, prefix)
    if tot is not None:
        out[key] = tot
        out[key + "_detail"] = det
json.dump(out, open(os.path.join(ROOT, "profiles", "emit_traffic.json"), "w"), indent=1)
print({k: v for k, v in out.items() if not k.endswith("_detail") and k not in ("source", "note")})
