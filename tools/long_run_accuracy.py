#!/usr/bin/env python3
"""Accuracy half of the metric at full cfg3 size, past burn-in: 256 x 256 particles, D = 32, N = 1e5, reference defaults
(burn-in 1000), 3000 iterations; posterior mean and spread of the last 1000 iterations against the closed-form
conjugate posterior.  SUFFSTAT likelihood (same posterior, 0.04 ms/iteration) by default, `streaming` as argument
for the reference-faithful evaluation (17 s).  History stays on the device (50 GB of the 288 GB) and comes back in
slices."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demc_amd import workloads as W  # noqa: E402
import demc_amd  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "suffstat"
G, Np, N, d, iters, keep = 256, 256, 100000, 32, 3000, 1000
P = G * Np
prob = W.cfg3(N=N, d=d, G=G, Np=Np)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else iters
kw = dict(n_groups=G, Np=Np, D=d, schedule=2, seed=20260001, burnin=1000, loglike_mode=0 if mode == "streaming" else 1, trace=0)
# stage 1: everything before the kept window, no history; stage 2: the kept window with history
pre = demc_amd.HipEngine(n_rows=0, store_history=0, **kw)
W.configure(pre, prob)
pre.set_state(prob["init"](P, np.random.default_rng(20260003)))
t0 = time.perf_counter()
pre.step(1, iters - keep)
state = pre.get_state()
pre.close()
eng = demc_amd.HipEngine(n_rows=iters, **kw) if iters <= 3000 else None
if eng is None:  # long runs: history rows are indexed by iteration, so the window is replayed at iterations 1001.. of a fresh handle
    eng = demc_amd.HipEngine(n_rows=1000 + keep, **kw)
    first = 1001
else:
    first = iters - keep + 1
W.configure(eng, prob)
eng.set_state(*state)
eng.step(first, keep)
dt = time.perf_counter() - t0
iters_hist0 = first - 1
s1 = np.zeros(d); s2 = np.zeros(d); n = 0; acc = 0.0
for r0 in range(iters_hist0, iters_hist0 + keep, 50):
    th, a, _, _ = eng.get_history(r0, r0 + 50)
    x = th.reshape(-1, d)
    s1 += x.sum(0); s2 += (x * x).sum(0); n += x.shape[0]; acc += a.mean() * 50
eng.close()
mean = s1 / n
sd = np.sqrt(s2 / n - mean * mean)
Ainv = np.linalg.inv(prob["hyper"])
prec = N * Ainv + np.eye(d)
post_mean = np.linalg.solve(prec, N * Ainv @ prob["data"].mean(0))
post_sd = np.sqrt(np.diag(np.linalg.inv(prec)))
print(json.dumps(dict(mode=mode, iterations=iters, kept=keep, seconds=dt,
                      posterior_mean_l1_rel=float(np.abs(mean - post_mean).sum() / np.abs(post_mean).sum()),
                      max_abs_err_in_posterior_sd=float(np.max(np.abs(mean - post_mean) / post_sd)),
                      sd_ratio_min=float((sd / post_sd).min()), sd_ratio_median=float(np.median(sd / post_sd)),
                      sd_ratio_max=float((sd / post_sd).max()), accept_rate=float(acc / keep))))
