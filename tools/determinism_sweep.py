#!/usr/bin/env python3
"""Race hunt: every random configuration (families x shapes x sampler settings x schedules x likelihood / fuse modes) is run
several times on fresh handles and the complete outputs -- state, slot-keyed history, device chain export -- must agree bit
for bit.  A kernel-level or stream-ordering race shows up as a run that differs from its siblings.

    python3 tools/determinism_sweep.py [n_cases] [repeats] [seed]
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import demc_amd  # noqa: E402
from conftest import make_problem, setup_engine  # noqa: E402
from test_gpu_edge_cases import _fuzz_cases  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
bad = 0
for case in _fuzz_cases(n_cases, seed):
    fam, kw, cfg = case.values
    cfg = dict(cfg)
    n_iter = 25
    rng0 = np.random.default_rng(cfg["seed"])
    prob = make_problem(fam, rng0, **kw)
    th0 = prob["init"](cfg["n_groups"] * cfg["Np"])
    for fuse in (0, 2, 1):
        sigs = set()
        for r in range(repeats):
            e = demc_amd.HipEngine(D=prob["D"], n_rows=n_iter, trace=0, fuse=fuse, **cfg)
            setup_engine(e, prob)
            e.set_state(th0)
            e.step(1, n_iter)
            parts = list(e.get_state()) + list(e.get_history(0, n_iter)) + [e.export_chains(0, n_iter)]
            e.close()
            sigs.add(hashlib.md5(b"".join(np.ascontiguousarray(p).tobytes() for p in parts)).hexdigest())
        if len(sigs) != 1:
            bad += 1
            print("NON-DETERMINISTIC", case.id, "fuse", fuse, kw, cfg, flush=True)
print(f"{n_cases} configurations x 3 fuse modes x {repeats} runs: {bad} non-deterministic")
sys.exit(1 if bad else 0)
