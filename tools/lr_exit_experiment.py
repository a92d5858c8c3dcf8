#!/usr/bin/env python3
"""What the long-row kernel's launch time is made of: runs cfg4 on the A/B build (make EXPERIMENTS=1) with the kernel leaving
at successive points (DEMC_LR_EXIT = 1: at entry, 2: after the prologue, 3: after the span loops, 4: after the rounds at the
edges, 5: before the row moves, 0: whole kernel) and prints the device time per launch (HIP events).  The shortened kernels
compute nothing useful: read the differences, never the absolute rates."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "csrc")
subprocess.check_call(["make", "-C", csrc, "-s", "EXPERIMENTS=1", "OUT=../libdemc_hip_exp.so"])
if len(sys.argv) > 1 and sys.argv[1] != "all":  # child: one exit point
    import numpy as np
    import demc_amd
    demc_amd._ffi.LIB_PATH = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "libdemc_hip_exp.so")
    from demc_amd import workloads as W
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    w = W.cfg4(G=G)
    eng = demc_amd.HipEngine(n_groups=w["G"], Np=w["Np"], D=w["D"], n_rows=64, schedule=2, seed=1, **w["engine"])
    W.configure(eng, w)
    eng.set_state(w["init"](w["G"] * w["Np"], np.random.default_rng(4)))
    eng.step(1, 10)
    eng.timing_enable(True)
    eng.step(11, 40)
    tm = eng.timing_read()
    print(f"exit {sys.argv[1]}: {tm['propose']['ms'] / tm['propose']['launches'] * 1e3:8.2f} us per launch "
          f"({tm['propose']['launches']} launches)")
    sys.exit(0)
for e in ("1", "2", "3", "4", "5", "0"):
    subprocess.check_call([sys.executable, __file__, e] + sys.argv[2:], env=dict(os.environ, DEMC_LR_EXIT=e))
