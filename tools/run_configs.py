#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the headline one on ONE GPU (their multi-GPU forms shard groups, so a
single rank's share is what one GPU does).  Prints per-kernel-class device time per iteration (HIP events)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import demc_amd as D  # noqa: E402
from demc_amd import families as F  # noqa: E402


def build(name, rng):
    inf = np.inf
    if name == "cfg1":  # Gaussian_Example.jl: D=2, N=50, 4 groups x 10
        data = rng.normal(0, 1, 50)
        return dict(G=4, Np=10, D=2, fam=F.FAM_GAUSSIAN, data=data, dims=[50], hyper=None, pk=[1, 2], pa=[0, 0], pb=[1, 1],
                    pref=[0, 0], lo=[-inf, 0], hi=[inf, inf], extra={},
                    init=lambda P: np.stack([rng.normal(0, 1, P), np.abs(rng.standard_cauchy(P)) + .1], 1))
    if name == "cfg2":  # MvNormal D=8, 32 x 64, N=1e4
        d, N = 8, 10000
        A = rng.normal(0, 1, (d, d)); S = A @ A.T / d + 0.5 * np.eye(d)
        X = rng.normal(0, 1, d) + rng.normal(0, 1, (N, d)) @ np.linalg.cholesky(S).T
        return dict(G=32, Np=64, D=d, fam=F.FAM_MVN_FULL, data=X, dims=[N, d], hyper=S, pk=[1] * d, pa=[0] * d, pb=[1] * d,
                    pref=[0] * d, lo=[-inf] * d, hi=[inf] * d, extra={}, init=lambda P: rng.normal(0, 1, (P, d)))
    if name == "cfg4":  # hierarchical Binomial, S=1e4 subjects, one GPU's share of 128 groups = 16 groups x 32
        S_, n = 10000, 50.0
        b0 = rng.normal(0, 1, S_)
        k = rng.binomial(50, 1 / (1 + np.exp(-(1 + b0)))).astype(float)
        Dd = S_ + 2
        m0 = np.zeros(Dd, np.uint8); m0[:2] = 1
        return dict(G=16, Np=32, D=Dd, fam=F.FAM_HIER_BINOMIAL, data=k, dims=[S_], hyper=[n], pk=[1, 2] + [5] * S_,
                    pa=[1, 0] + [0] * S_, pb=[1, 1] + [1] * S_, pref=[0, 0] + [1] * S_, lo=[-inf, 0] + [-inf] * S_, hi=[inf] * Dd,
                    extra=dict(masks=np.stack([m0, 1 - m0])),
                    init=lambda P: np.concatenate([rng.normal(1, 1, (P, 1)), np.abs(rng.standard_cauchy((P, 1))) + .3,
                                                   rng.normal(0, 1, (P, S_))], 1))
    if name == "cfg5":  # LBA 3 accumulators, 5e4 trials, one GPU's share of 512 groups = 64 groups x 128, snooker on
        N, na = 50000, 3
        choice = rng.integers(1, na + 1, N).astype(float); rt = rng.uniform(0.45, 1.6, N); mr = rt.min()
        Dd = na + 3
        return dict(G=64, Np=128, D=Dd, fam=F.FAM_LBA, data=np.concatenate([choice, rt]), dims=[N, na], hyper=None,
                    pk=[1] * na + [1, 1, 3], pa=[1] * na + [.8, .2, 0.], pb=[5] * na + [.2, .1, mr], pref=[0] * Dd, lo=[0] * Dd,
                    hi=[inf] * (Dd - 1) + [mr], extra=dict(theta_snooker=0.1),
                    init=lambda P: np.concatenate([rng.uniform(.5, 4, (P, na)), rng.uniform(.5, 1.1, (P, 1)),
                                                   rng.uniform(.05, .4, (P, 1)), rng.uniform(.05, mr * .9, (P, 1))], 1))
    raise KeyError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["cfg1", "cfg2", "cfg4", "cfg5"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--mode", default="streaming")
    a = ap.parse_args()
    for name in a.configs:
        rng = np.random.default_rng(20260000 + int(name[-1]))
        c = build(name, rng)
        P = c["G"] * c["Np"]
        ex = dict(c["extra"])
        masks = ex.pop("masks", None)
        eng = D.HipEngine(n_groups=c["G"], Np=c["Np"], D=c["D"], n_rows=a.steps * 2 + 2, schedule=2, seed=1,
                          loglike_mode=0 if a.mode == "streaming" else 1, **ex)
        eng.set_model(c["fam"], c["data"], c["dims"], c["hyper"])
        eng.set_priors(c["pk"], c["pa"], c["pb"], c["pref"])
        eng.set_bounds(c["lo"], c["hi"])
        if masks is not None:
            eng.set_blocks(masks)
        eng.set_state(c["init"](P))
        eng.step(1, 2)
        t0 = time.perf_counter()
        eng.step(3, a.steps)
        dt = time.perf_counter() - t0
        eng.timing_enable(True)
        eng.step(3 + a.steps, a.steps)
        tm = eng.timing_read()
        eng.timing_enable(False)
        acc = eng.get_history(2, 2 + a.steps)[1].mean()
        eng.close()
        print(json.dumps(dict(config=name, P=P, D=c["D"], ms_per_iter=dt / a.steps * 1e3, particle_updates_per_s=P * a.steps / dt,
                              accept_rate=float(acc), device_ms_per_iter={k: v["ms"] / a.steps for k, v in tm.items()})))


if __name__ == "__main__":
    main()
