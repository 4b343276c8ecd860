import os, subprocess, sys, re
libs = sys.argv[1:]
cfgs = {"C2": ["--batch", "4"], "C2b1": ["--batch", "1"], "C5": ["--n", "200000", "--half", "100", "--P", "30000", "--batch", "4", "--iters", "100"], "C5b1": ["--n", "200000", "--half", "100", "--P", "30000", "--batch", "1", "--iters", "100"]}
for rnd in range(2):
    for l in libs:
        for name, extra in cfgs.items():
            out = subprocess.run([sys.executable, "tools/bench_vox.py"] + extra, env=dict(os.environ, PP_HIP_LIB=os.path.abspath(l)), capture_output=True, text=True).stdout
            m = re.search(r"([\d.]+) us/step.*emit median ([\d.]+) us", out)
            print(l, name, "step", m.group(1), "emit", m.group(2), flush=True)
