#!/usr/bin/env python3
"""Where K1 spends its cycles: runs a workload on the DIAGNOSTIC build of the library (in-kernel s_memtime stamps,
`make -C differentialevolutionmcmc.jl_amd/csrc STAMPS=1 OUT=../libdemc_hip_stamps.so`) and prints, per workgroup,
the median cycle count at each stamp of the fused propose kernel.  Never quote this build's run time: read shares."""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "csrc")
if os.environ.get("STAMPS_PREBUILT") != "1":  # (build here and ship the library to the GPU box: STAMPS_PREBUILT=1 skips the make)
  subprocess.check_call(["make", "-B", "-j8", "-C", csrc, "-s", "STAMPS=1", "OUT=../libdemc_hip_stamps.so"] +
                        (["STAMP_PASS=" + os.environ["STAMP_PASS"]] if "STAMP_PASS" in os.environ else []) +
                        (["EXTRA=" + os.environ["STAMP_EXTRA"]] if "STAMP_EXTRA" in os.environ else []))  # e.g. -DDEMC_X_...=1 experiments
import demc_amd  # noqa: E402
demc_amd._ffi.LIB_PATH = os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "libdemc_hip_stamps.so")
from demc_amd import workloads as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-groups", type=int, default=None)
ap.add_argument("--np", type=int, default=256, dest="Np")
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--nobs", type=int, default=100000)
ap.add_argument("--mode", default="suffstat")
ap.add_argument("--config", default="")
ap.add_argument("--partners", default="current", choices=["current", "history"],
                help="history: DE-MC_Z (`resample`), synchronous schedule, 16 prior rows, past burn-in: ONE k_propose launch per iteration")
a = ap.parse_args()
if a.config:  # one of tools/run_configs.py's BASELINE shapes (e.g. cfg4: a whole workgroup per particle)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import run_configs  # noqa: E402
    c = run_configs.build(a.config, np.random.default_rng(20260000 + int(a.config[-1])), **({"G": a.n_groups} if a.n_groups else {}))
    ex = dict(c["extra"])
    masks = ex.pop("masks", None)
    eng = demc_amd.HipEngine(n_groups=c["G"], Np=c["Np"], D=c["D"], n_rows=40, schedule=2, seed=1, trace=0, **ex)
    eng.set_model(c["fam"], c["data"], c["dims"], c["hyper"])
    eng.set_priors(c["pk"], c["pa"], c["pb"], c["pref"])
    eng.set_bounds(c["lo"], c["hi"])
    if masks is not None:
        eng.set_blocks(masks)
    eng.set_state(c["init"](c["G"] * c["Np"]))
    eng.step(1, 30)
    if masks is not None and os.environ.get("STAMP_BLOCK"):  # stamps of ONE block sweep's launch (e.g. 0: the hyper-parameter sweep)
        eng.set_blocks(masks[int(os.environ["STAMP_BLOCK"]):int(os.environ["STAMP_BLOCK"]) + 1])
        eng.step(int(os.environ.get("STAMP_ITER", "31")), 2)  # (an iteration past n_rows = 40: no history row is due -- the LITE instance)
    n_wg = c["G"] * c["Np"] // 2  # upper bound on the workgroups of a launch; unused stamp slots are filtered out below
else:
    a.n_groups = a.n_groups or 256
    prob = W.cfg3(N=a.nobs, d=a.dim, G=a.n_groups, Np=a.Np)
    hist = a.partners == "history"
    eng = demc_amd.HipEngine(n_groups=a.n_groups, Np=a.Np, D=a.dim, n_rows=60 if hist else 40, schedule=1 if hist else 2, seed=1,
                             loglike_mode=1 if a.mode == "suffstat" else 2 if a.mode == "direct" else 0, trace=0,
                             **(dict(partner_kind=1, n_initial=16, burnin=0) if hist else {}))
    W.configure(eng, prob)
    rng0 = np.random.default_rng(20260003)
    if hist:
        eng.set_history_rows(0, np.stack([prob["init"](a.n_groups * a.Np, rng0) for _ in range(16)]))
    eng.set_state(prob["init"](a.n_groups * a.Np, rng0))
    eng.step(17 if hist else 1, 30)
    n_wg = min(512, (a.n_groups * a.Np) // 16) if (os.environ.get('DEMC_RESIDENT') == '0' or hist) else a.n_groups  # resident: one per group
if os.environ.get("STAMP_NWG"):  # (e.g. 256: k_res_mvn<..., HIST> is one workgroup per group; slots beyond hold an earlier kernel's stamps)
    n_wg = int(os.environ["STAMP_NWG"])
w_prop = eng.get_trace()["w_prop"]
if os.environ.get("STAMP_TIMELINE") == "1":  # STAMP_EXTRA=-DDEMC_STAMPS_TIMELINE: (start, end) of every particle, 100 MHz ticks
    tl = w_prop[: 2 * (len(w_prop) // 2)].reshape(-1, 2)
    idx = np.nonzero(tl[:, 1] > 0)[0]
    tl = tl[idx]
    mut = (tl[:, 1] % 1.0) > 0.25
    t0 = tl[:, 0].min()
    st, en = (tl[:, 0] - t0) * 0.01, (np.floor(tl[:, 1]) - t0) * 0.01
    grid = int((st < 1.0).sum())  # persistent kernel: the workgroups all start with the launch, particle vb runs in workgroup vb % grid
    print(f"{len(tl)} particles ({int(mut.sum())} in a mutating group) in {grid} workgroups; launch ends {en.max():.1f} us after the first start")
    for pos in range((len(tl) + grid - 1) // grid):
        sel = (idx // grid) == pos
        if not sel.any():
            continue
        d = (en - st)[sel]
        print(f"  particle {pos + 1} of a workgroup: starts {np.percentile(st[sel], 10):6.1f} .. {np.percentile(st[sel], 90):6.1f} us (p10 .. p90), "
              f"takes {np.median(d):5.1f} us (p10 {np.percentile(d, 10):.1f}, p90 {np.percentile(d, 90):.1f}), ends by {en[sel].max():6.1f}")
    print("  mutation particles take %.1f us (median), the others %.1f" % (np.median((en - st)[mut]) if mut.any() else float("nan"),
                                                                          np.median((en - st)[~mut])))
    sys.exit(0)
if a.mode in ("streaming", "direct"):
    n_wg = len(w_prop) // 24  # streaming-resident form: several workgroups per group
n_wg = min(n_wg, len(w_prop) // 24)  # the stamps live in the P-long trace array, 24 per workgroup
full = w_prop[: n_wg * 24].reshape(n_wg, 24)
full = full[full[:, 10] > 0]
t = full[:, :11]
names = ["softmax prefix sums done (tile in flight)", "plan written; tile landed (LDS-DMA wait + barrier)",
         "top of steady-state pass (passes before it)", "particle Philox blocks + broadcast", "indices / gammas / base pick",
         "per-dimension loop (noise, proposal, bounds, prior)", "sub-group reductions", "MvNormal preparation",
         "in-kernel observation loop", "accept + row moves", "kernel end (remaining passes)"]
if a.config:  # k_longrow (demc_longrow.hpp): wave 1's stamps, a typical wave (wave 0 also picks the base)
    m = np.median(full, 0)
    print(f"{len(full)} workgroups of k_longrow; cycles since kernel start (median over workgroups), wave 1 unless noted")
    for label, v in (("kernarg in, addresses formed", m[16]), ("Philox block of the per-particle scalars", m[17]),
                     ("per-particle scalars drawn", m[12]), ("wave 0: base picked", m[0]), ("past the first barrier", m[1]),
                     ("wave 0: hyper-parameter scalars proposed", m[4]), ("spans done", m[15]), ("... by wave 0", m[13]), ("... by the last wave", m[14]),
                     ("rounds at the edges done (one scalar per lane)", m[5]), ("reductions done (slowest wave in)", m[6]), ("accept + row moves done", m[9])):
        print(f"  {label:45s} {v:9.0f}")
    print(f"  blocks of lane 64 in the span loops: {m[2]:.0f}")
    if m[18] > 0:  # s_memrealtime runs at 100 MHz: the shader clock while the kernel ran, and when the workgroups started
        print(f"  shader clock of the run: {m[10] / m[18] * 0.1:.2f} GHz (kernel end {m[10]:.0f} cycles = {m[18] * 0.01:.1f} us)")
        st = np.sort(full[:, 19] - full[:, 19].min()) * 0.01
        print("  workgroup starts, us since the first:", " ".join(f"{v:.0f}" for v in st))
    print("  accept + row moves done, per workgroup:", " ".join(f"{v:.0f}" for v in full[:, 9]))
    print("  spans done, per workgroup:", " ".join(f"{v:.0f}" for v in full[:, 15]))
    sys.exit(0)
med = np.median(t, 0)
prev = 0.0
ran = eng.last_kernels()
if ran.startswith("k_res_mvn"):
    # the lean kernel (demc_resmvn.hpp) stamps the LAST colour phase (DE-MC_Z: the second half) relative to that phase's start, in
    # the slots 0, 1, 4, 5, 7, 9, 10; the other slots still hold what an earlier kernel of the run left there
    print(f"{len(t)} workgroups of {ran}; cycles since the start of the last phase (median over workgroups), and the step")
    for label, slot in (("wave 0: select_base's cumulative weights in LDS", 0), ("PART and NOISE blocks drawn", 1),
                        ("base picked (DE-MC_Z: partner cells found)", 4), ("proposal, bounds, prior of the lane's scalars", 5),
                        ("A^-1 product and its dot products (STREAM: + cross terms handed over: slot 8)", 7),
                        ("accept + row moves", 9), ("end of the phase (after its barrier)", 10)):
        print(f"  {label:80s} {med[slot]:9.0f}  (+{med[slot] - prev:7.0f})")
        prev = med[slot]
    if a.mode not in ("streaming", "direct"):
        sys.exit(0)
    prev = 0.0
print(f"{len(t)} workgroups; cycles since kernel start (median), and the step")
for n, m in zip(names, med):
    print(f"  {n:55s} {m:9.0f}  (+{m - prev:7.0f})")
    prev = m
ex = np.median(full[:, [11, 13]], 0)
print(f"  also: plan written (before the tile wait) {ex[0]:.0f}; A^-1 fragments in registers (before pass 0) {ex[1]:.0f}")
pro = np.median(full[:, [12, 14, 15]], 0)
print(f"  inside the prologue: group coin {pro[0]:.0f}; weights / A^-1 parked {pro[1]:.0f}; tile copy issued {pro[2]:.0f}")
if a.mode in ("streaming", "direct") and (full[:, 16] > 0).any():
    st = np.median(full[full[:, 16] > 0][:, 16:21], 0)
    print("  streaming-resident: all proposals prepared %.0f; chunk cross terms %.0f; granules stored %.0f; collected %.0f; "
          "accept + moves done %.0f" % tuple(st))
    if a.mode == "direct":  # (the DIRECT instance: its residual stage from the inside -- slots 20, 21)
        print("  DIRECT stage: m in registers %.0f; residual loop done %.0f; row partials in LDS %.0f" % (tuple(np.median(full[full[:, 16] > 0][:, 20:22], 0)) + (np.median(full[full[:, 16] > 0][:, 15]),)))
    if (full[:, 23] > 0).any():  # k_res_mvn: store / collect times of the workgroups of a group on one clock (slots 22, 23)
        # (blockIdx = j * 8 + xcd; workgroup c of group g = 8 * (j / C) + xcd sits at j = (g / 8) * C + c; the trace holds the first P / 24 blocks)
        C = 8
        blk = np.arange(full.shape[0])
        grp = ((blk >> 3) // C) * 8 + (blk & 7)
        rows = []
        for g_ in np.unique(grp):
            m_ = grp == g_
            if m_.sum() != C:
                continue
            st_, co_ = full[m_, 22], full[m_, 23]
            rows.append((st_.max() - st_.min(), co_.min() - st_.max(), co_.max() - st_.max()))
        cpos = (blk >> 3) % C
        whole = np.isin(grp, [g_ for g_ in np.unique(grp) if (grp == g_).sum() == C])
        first = {g_: full[grp == g_, 22].min() for g_ in np.unique(grp[whole])}
        rel = np.array([full[i, 22] - first[grp[i]] for i in np.flatnonzero(whole)])
        relc = np.array([full[i, 23] - first[grp[i]] for i in np.flatnonzero(whole)])
        print("  store time after the group's first store, by workgroup c (median over groups, 10 ns ticks): " +
              " ".join("%.0f" % np.median(rel[cpos[whole] == c_]) for c_ in range(C)) + ";  collect: " +
              " ".join("%.0f" % np.median(relc[cpos[whole] == c_]) for c_ in range(C)))
        r_ = np.median(np.array(rows), 0)
        print("  hand-over on the 100 MHz clock (10 ns ticks), %d whole groups (median): last store - first store %.0f; first collect - last store %.0f; "
              "last collect - last store %.0f" % ((len(rows),) + tuple(r_)))
    if (full[:, 11] > 0).any():  # k_res_mvn: the next phase's draws sit between the store and the first poll
        print("  next phase's blocks drawn (between store and poll) %.0f" % np.median(full[full[:, 16] > 0][:, 11]))
