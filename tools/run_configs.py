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


def build(name, rng, **kw):
    """the BASELINE configs as demc_amd.workloads builds them (bench.py's data), in this tool's older dict shape"""
    from demc_amd import workloads as W
    w = W.BUILDERS[name](**kw)
    extra = dict(w["engine"])
    if w["masks"] is not None:
        extra["masks"] = w["masks"]
    return dict(G=w["G"], Np=w["Np"], D=w["D"], fam=w["fam"], data=w["data"], dims=w["dims"], hyper=w["hyper"], pk=w["pk"], pa=w["pa"],
                pb=w["pb"], pref=w["pref"], lo=w["lo"], hi=w["hi"], extra=extra, init=lambda P: w["init"](P, rng))


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
