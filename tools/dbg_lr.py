import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import demc_amd
from oracle import oracle as O
from conftest import make_problem, setup_engine
prob = make_problem("hier_binomial", np.random.default_rng(7), S=3000)
D = prob["D"]
m0 = np.zeros(D, np.uint8); m0[:2] = 1
for masks in (np.stack([m0]), np.stack([1 - m0]), None):
    cfg = dict(n_groups=2, Np=6, D=D, n_rows=2, schedule=2, seed=5, burnin=3, n_blocks=0 if masks is None else 1)
    eng = demc_amd.HipEngine(**cfg); o = O.Oracle(**{k: v for k, v in cfg.items() if k in O.CFG_KEYS})
    for e in (eng, o):
        setup_engine(e, prob)
        if masks is not None: e.set_blocks(masks)
    th0 = prob["init"](12)
    eng.set_state(th0); th, w, ids = eng.get_state(); o.set_state(th, w, ids)
    eng.step(1, 1); o.step(1, 1)
    tg, to = eng.get_trace(), o.get_trace()
    print("mask", None if masks is None else masks[0][:4], "idx equal", np.array_equal(tg["idx"], to["idx"]))
    print(" w_prop gpu", tg["w_prop"]); print(" w_prop orc", to["w_prop"])
    print(" acc", tg["accepted"], to["accepted"], "prop maxdiff", np.abs(tg["proposal"] - to["proposal"]).max())
    eng.close(); o.close()
