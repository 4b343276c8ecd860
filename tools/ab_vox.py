"""Interleaved A/B of two libpp_hip builds in one process tree (dev aid)."""
import os, subprocess, sys, re, statistics
libs = sys.argv[1:]
res = {l: {1: [], 4: []} for l in libs}
for rnd in range(3):
    for l in libs:
        for B in (1, 4):
            out = subprocess.run([sys.executable, "tools/bench_vox.py", "--batch", str(B), "--iters", "300"],
                                 env=dict(os.environ, PP_HIP_LIB=os.path.abspath(l)), capture_output=True, text=True).stdout
            m = re.search(r"([\d.]+) us/step.*emit median ([\d.]+) us", out)
            res[l][B].append((float(m.group(1)), float(m.group(2))))
for l in libs:
    for B in (1, 4):
        print(l, "B=%d" % B, "step us", [r[0] for r in res[l][B]], "emit us", [r[1] for r in res[l][B]])
