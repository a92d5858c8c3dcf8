#!/usr/bin/env python3
"""Where a launch of k_frozen_sweep spends its time: runs cfg4 on the DIAGNOSTIC build of the library (in-kernel stamps:
`make -C differentialevolutionmcmc.jl_amd/csrc STAMPS=1 OUT=../libdemc_hip_stamps.so`), then ONE block sweep alone, and prints
(a) when the launch's workgroups start and end (100 MHz clock: the generations of workgroups on the CUs), (b) wave 0's shader
cycles at the kernel's stages (median over the first P / 8 workgroups).  Never quote this build's run time: read shares.
    python3 tools/frozen_stamps.py [block] [n_groups] [burnin]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import demc_amd  # noqa: E402
demc_amd._ffi.LIB_PATH = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "libdemc_hip_stamps.so")
import run_configs  # noqa: E402

block = int(sys.argv[1]) if len(sys.argv) > 1 else 0
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
burnin = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
c = run_configs.build("cfg4", np.random.default_rng(20260004), G=G)
ex = dict(c["extra"])
masks = ex.pop("masks")
ex["burnin"] = burnin
eng = demc_amd.HipEngine(n_groups=c["G"], Np=c["Np"], D=c["D"], n_rows=40, schedule=2, seed=1, trace=0, **ex)
eng.set_model(c["fam"], c["data"], c["dims"], c["hyper"])
eng.set_priors(c["pk"], c["pa"], c["pb"], c["pref"])
eng.set_bounds(c["lo"], c["hi"])
eng.set_blocks(masks)
eng.set_state(c["init"](c["G"] * c["Np"]))
eng.step(1, 30)
eng.set_blocks(masks[block:block + 1])
eng.step(41, 1)  # (an iteration past n_rows: no history row is due)
print("kernel:", eng.last_kernels())
tr = eng.get_trace()
tl = tr["log_adj"].reshape(-1, 2)
n = c["G"] * c["Np"] // 2
tl = tl[:n]
t0 = tl[:, 0].min()
st, en = (tl[:, 0] - t0) * 0.01, (np.floor(tl[:, 1]) - t0) * 0.01
code = np.round((tl[:, 1] % 1.0) * 8).astype(int)
kind, acc = code // 2, code % 2
print(f"{n} workgroups of the LAST phase; launch ends {en.max():.1f} us after the first start; accepted {acc.mean():.3f}; "
      f"kinds {np.bincount(kind, minlength=3)}")
h, edges = np.histogram(st, bins=np.arange(0, en.max() + 5, 5.0))
print("  starts per 5 us:", " ".join(str(v) for v in h))
h, _ = np.histogram(en, bins=edges)
print("  ends   per 5 us:", " ".join(str(v) for v in h))
d = en - st
for k in range(3):
    if (kind == k).any():
        sel = kind == k
        print(f"  kind {k}: {sel.sum()} workgroups take {np.median(d[sel]):.1f} us (p10 {np.percentile(d[sel], 10):.1f}, p90 {np.percentile(d[sel], 90):.1f}); "
              f"accepted ones {np.median(d[sel & (acc == 1)]) if (sel & (acc == 1)).any() else float('nan'):.1f}")
first = st < 1.0
print(f"  workgroups starting in the first us: {first.sum()}; they take {np.median(d[first]):.1f} us; the later ones {np.median(d[~first]) if (~first).any() else float('nan'):.1f}")
w = tr["w_prop"][: (len(tr["w_prop"]) // 8) * 8].reshape(-1, 8)[: min(n, len(tr["w_prop"]) // 8)]
w = w[w[:, 6] > 0]
names = ["per-particle scalars, partner rows known", "base picked (barrier)", "snooker projections", "theta'[0], reference scalars known (barrier)",
         "the pass (wave 0)", "reduced and decided (barrier)", "row moves, end"]
for kk in range(3):
    sel = kind[: len(w)] == kk
    if not sel.any():
        continue
    m = np.median(w[sel], 0)
    print(f"  kind {kk}: shader cycles since the workgroup's start (median over {sel.sum()} workgroups), and the step")
    prev = 0.0
    for i, nm in enumerate(names):
        print(f"    {nm:50s} {m[i]:9.0f}  (+{m[i] - prev:7.0f})")
        prev = m[i]
