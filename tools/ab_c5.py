import os, subprocess, sys, re
libs = sys.argv[1:]
for rnd in range(2):
    for l in libs:
        out = subprocess.run([sys.executable, "tools/bench_vox.py", "--n", "200000", "--half", "100", "--P", "30000", "--batch", "1", "--iters", "100"], env=dict(os.environ, PP_HIP_LIB=os.path.abspath(l)), capture_output=True, text=True).stdout
        m = re.search(r"([\d.]+) us/step.*emit median ([\d.]+) us", out)
        print(l, "C5b1 step", m.group(1), "emit", m.group(2), flush=True)
