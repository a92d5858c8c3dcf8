#!/usr/bin/env python3
"""Counts what ONE (trial, proposal) evaluation of the LBA likelihood executes, from the compiler's own output: the device
code of k_lba_wave<3> (the wave-per-proposal kernel that ships; round 4 / early round 5: k_obs_loglike) is compiled to assembly,
its batch loop (kLbaBatch trials per lane and iteration) is cut out, and its FP64 instructions are counted -- an FMA as two flop, add / mul / max / min / rcp as one.  bench.py's cfg5
roofline uses the result (profiles/<round>/lba_inner_loop.json) instead of a hand count; the PMC pass of the same round gives
the executed VALU instructions per evaluation to compare with `valu_insts`.

    python3 tools/count_lba_flop.py profiles/r03/lba_inner_loop.json"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = int(re.search(r"constexpr int kLbaBatch = (\d+);", open(os.path.join(ROOT, "differentialevolutionmcmc.jl_amd/csrc/demc_kernels.hpp")).read()).group(1))
DEG = int(re.search(r"constexpr int kPhiDeg = (\d+);", open(os.path.join(ROOT, "differentialevolutionmcmc.jl_amd/csrc/demc_phi_table.hpp")).read()).group(1))
READS = (DEG + 2) // 2  # sixteen-byte LDS reads per table look-up: deg + 1 coefficients, two per read
src = ('#include "%s"\nnamespace demc { template __global__ void k_lba_wave<3>(KParams, int, unsigned long long*); }\n'
       % os.path.join(ROOT, "differentialevolutionmcmc.jl_amd/csrc/demc_kernels.hpp"))
with tempfile.TemporaryDirectory() as td:
    hip, asm = os.path.join(td, "k.hip"), os.path.join(td, "k.s")
    open(hip, "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", hip, "-o", asm], stderr=subprocess.DEVNULL)
    text = open(asm).read()
body = re.search(r"^_ZN4demc10k_lba_waveILi3EEEvNS_7KParamsEiPy:(.*?)s_endpgm", text, re.S | re.M).group(1)
# the batch loop of the 3-accumulator instance: the loop (header label .. last branch back to it) whose body holds exactly
# 3 accumulators x 2 look-ups x READS sixteen-byte table reads per trial
lines = body.splitlines()
labels = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
loops = {}
for lab, start in labels.items():
    back = [i for i, ln in enumerate(lines) if i > start and re.search(r"s_c?branch\w* " + re.escape(lab) + r"$", ln.strip())]
    if back:
        loops[lab] = (start, back[-1] + 1)
best = None
for lab, (a, b) in loops.items():
    # the cold path of the batch (a product that left the double range is redone with a log per trial: an inner loop of one
    # trial per trip, compiled inside the batch loop) is not part of what an evaluation executes: cut nested loops out
    inner = [(x, y) for l2, (x, y) in loops.items() if l2 != lab and x > a and y <= b]
    blk = [ln for i, ln in enumerate(lines[a:b], start=a) if not any(x <= i < y for x, y in inner)]
    if sum(x.strip().startswith("ds_read_b128") for x in blk) == 3 * 2 * READS * BATCH and (best is None or len(blk) < len(best)):
        best = blk
assert best, "batch loop not found"
ins = [x.split()[0] for x in best if x.startswith("\t") and not x.strip().startswith((";", "."))]
valu = [x for x in ins if x.startswith("v_")]
fma = sum(x.startswith(("v_fma_f64", "v_fmac_f64")) for x in valu)
one = sum(x.startswith(("v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64", "v_rcp_f64")) for x in valu)
lds = sum(x.startswith("ds_read") for x in ins)
# the log of the batch product is outside the counted set only if it is a call; here it is inline: count it with the loop
out = dict(kernel="k_lba_wave<3> batch loop", trials_per_iteration=BATCH,
           valu_insts_per_eval=len(valu) / BATCH, fp64_fma_per_eval=fma / BATCH, fp64_other_per_eval=one / BATCH,
           fp64_flop_per_eval=(2 * fma + one) / BATCH, lds_reads_per_eval=lds / BATCH, salu_per_eval=sum(x.startswith("s_") for x in ins) / BATCH,
           note="static count over one iteration of the batch loop (both the winner's and the loser's factor of every accumulator "
                "are in the loop body -- a lane takes one of them -- so the per-evaluation figures are upper bounds by ~4 instructions "
                "per accumulator)")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
out["source_sha16"] = bench.source_fingerprint()  # bench.py quotes the count only for the sources it was taken on
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
