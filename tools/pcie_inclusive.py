#!/usr/bin/env python3
"""Whole-job rate of the headline workload INCLUDING the host<->device copies the boundary implies: upload of the data
and the initial state (host buffers handed to demc_set_model / demc_set_state), the iterations, and the download of the
whole history through demc_export_chains.  bench.py's `value` excludes the copies (inputs resident); this is the
PCIe-inclusive figure quoted in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demc_amd import workloads as W  # noqa: E402
import demc_amd  # noqa: E402

G, Np, N, d, iters = 256, 256, 100000, 32, 250
mode = sys.argv[1] if len(sys.argv) > 1 else "direct"
P = G * Np
prob = W.cfg3(N=N, d=d, G=G, Np=Np)
th0 = prob["init"](P, np.random.default_rng(20260003))
eng = demc_amd.HipEngine(n_groups=G, Np=Np, D=d, n_rows=iters, schedule=2, seed=20260001, loglike_mode={"streaming": 0, "suffstat": 1, "direct": 2}[mode],
                         trace=0)
t0 = time.perf_counter()
W.configure(eng, prob)          # X upload (25.6 MB) + fragment reorder + priors
eng.set_state(th0)                     # theta upload (16.8 MB) + initial evaluation
t1 = time.perf_counter()
eng.step(1, iters)
t2 = time.perf_counter()
out = eng.export_chains(0, iters)   # [iters][D+2][P] doubles to the host
t3 = time.perf_counter()
print(json.dumps(dict(mode=mode, iterations=iters, upload_s=t1 - t0, iterate_s=t2 - t1, download_s=t3 - t2,
                      download_GB=out.nbytes / 1e9, updates_per_s_resident=P * iters / (t2 - t1),
                      updates_per_s_pcie_inclusive=P * iters / (t3 - t0))))
