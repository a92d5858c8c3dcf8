#!/usr/bin/env python3
"""Long form of the randomised free-run tests (tests/test_gpu_production.py): N seeded random configurations of one generator,
each run free on the GPU and on the CPU oracle from the same start, compared as the tests compare them (every accept decision
and particle id equal, theta to 1e-10, log-posteriors to 1e-9).  Prints the failures and which kernel instances the cases ran.

    python3 tests/free_run_sweep.py {de_mc_z | two_colour | long_row | de_mc_z_families | row_streaming | per_observation | direct_resident} [n_cases] [seed]

It lives under tests/ because it runs the CPU oracle, which is test infrastructure: nothing outside tests/, the smoke check and
bench.py's cpu_baseline leg may use it.

(round 4: 200 x de_mc_z, 200 x two_colour and 60 x long_row were clean)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import demc_amd  # noqa: E402
from demc_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_gpu_production as T  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "de_mc_z"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
O.build()
gen = {"de_mc_z": T._de_mc_z_cases, "two_colour": T._two_colour_cases, "long_row": T._long_row_cases,
       "de_mc_z_families": T._de_mc_z_family_cases, "row_streaming": T._row_streaming_cases,
       "per_observation": T._per_observation_cases, "direct_resident": T._direct_resident_cases}[which]
bad, kernels = 0, {}
for case in gen(n, seed=seed):
    c = dict(case.values[0])
    try:
        if which == "de_mc_z_families":
            ran = T.run_de_mc_z_family_case(demc_amd, O, c)
        elif which == "row_streaming":
            ran = T.run_row_streaming_case(demc_amd, O, c)
        elif which == "per_observation":
            ran = T.run_per_observation_case(demc_amd, O, c)
        elif which == "direct_resident":
            ran = T.run_direct_resident_case(demc_amd, O, c)
        elif which == "long_row":
            S, G, Np, hist, blocks = c.pop("S"), c.pop("G"), c.pop("Np"), c.pop("hist"), c.pop("blocks")
            w = W.cfg4(S=S, G=G, Np=Np)
            if not blocks:
                w["masks"] = None
            if hist:
                c.update(schedule=1, partner_kind=1, n_initial=3)
            ran = T.free_run(demc_amd, O, w, (3 if hist else 0) + 5, [], G, Np, theta_exact=False, **c)
        else:
            d, Np, G = c.pop("d"), c.pop("Np"), c.pop("G")
            w = W.cfg3(N=700, d=d, G=G, Np=Np)
            if which == "de_mc_z":
                ran = T.free_run(demc_amd, O, w, c["n_initial"] + 10, [], G, Np, theta_exact=False, schedule=1, partner_kind=1, **c)
            else:
                ran = T.free_run(demc_amd, O, w, 12, [], G, Np, theta_exact=False, **c)
        kernels[ran] = kernels.get(ran, 0) + 1
    except AssertionError as e:
        first = str(e).splitlines()[0][:200]
        if "nothing was accepted" in first:  # (a degenerate draw of the generator: the comparison would be vacuous)
            print("vacuous:", case.id, flush=True)
            continue
        bad += 1
        print("FAIL:", case.id, first, flush=True)
print(f"{which}: {n} cases, {bad} failures")
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]):
    print(f"  {v:4d}  {k}")
sys.exit(1 if bad else 0)
