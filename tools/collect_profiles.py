#!/usr/bin/env python3
"""Collects the rocprofv3 evidence behind bench.py's `roofline` object and DESIGN.md section 6, on the GPU box:

    python3 tools/collect_profiles.py gpurun_out/profiles_rNN          # then copy the summaries into profiles/rNN/

For each mode (streaming = default bench, suffstat) it runs `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline`
under `rocprofv3 --kernel-trace --stats` (kernel_stats csv), then twice more under `--kernel-trace --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled by the readers of the
json, see bench.py), and for streaming once more with the matrix-pipe counters.  Each program is started directly
after `--` (no shell / env hop).  The fused K1 in its resident form covers a varying number of iterations per launch,
so its traffic is also reduced to bytes per ITERATION: all of its launches together span warmup + steps + the bench's
20 roofline iterations."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP, ROOFLINE_ITERS = 20, 5, 20


def run(out_dir, tag, prof_args, bench_args):
    d = os.path.join(out_dir, tag)
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace"] + prof_args + ["--output-format", "csv", "-d", d, "--", "python3",
                                                        os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(out_dir, tag + ".log"), "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=True, timeout=600)
    return d


def counters(d, skip=2):
    """kernel -> counter -> list of per-dispatch values (instances summed), in dispatch order"""
    vals = defaultdict(lambda: defaultdict(list))
    grid = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per, order = defaultdict(float), []
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (int(row["Dispatch_Id"]), row["Kernel_Name"], row["Counter_Name"])
                if key not in per:
                    order.append(key)
                per[key] += float(row["Counter_Value"])
                grid[row["Kernel_Name"]] = int(row["Grid_Size"])
        for key in sorted(order):
            vals[key[1]][key[2]].append(per[key])
    return vals, grid


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    out_dir = os.path.abspath(sys.argv[1])
    os.makedirs(out_dir, exist_ok=True)
    base = ["--steps", str(STEPS), "--warmup", str(WARMUP), "--no-cpu-baseline"]
    for mode in ("streaming", "suffstat"):
        args = base + (["--mode", "suffstat"] if mode == "suffstat" else [])
        d = run(out_dir, f"{mode}_stats", ["--stats"], args)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(out_dir, f"bench_cfg3_{mode}_kernel_stats.csv"))
        res = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            vals, grid = counters(run(out_dir, f"{mode}_{ctr}", ["--pmc", ctr], args))
            for kname, cs in vals.items():
                v = cs[ctr]
                e = res.setdefault(short(kname), {"grid_size": grid[kname]})
                use = v[2:] if len(v) > 2 else v  # skip the first launches (initial evaluation, cold caches)
                e[ctr + "_KB_mean"] = sum(use) / len(use)
                e["launches_" + ctr] = len(use)
                e[ctr + "_KB_total"] = sum(v)
        # fused K1 (resident or per-phase): HBM bytes per iteration over the whole run
        fused = [k for k in res if "k_propose" in k and res[k].get("launches_FETCH_SIZE", 0) >= 1 and "false" not in k.split(",")[1]]
        if mode == "suffstat" and fused:
            n_it = STEPS + WARMUP + ROOFLINE_ITERS
            tot = sum(2.0 * res[k]["FETCH_SIZE_KB_total"] + res[k]["WRITE_SIZE_KB_total"] for k in fused) * 1024.0
            res["k_propose_fused_per_iteration"] = {"bytes": tot / n_it, "iterations": n_it, "kernels": fused,
                                                    "note": "2 x FETCH_SIZE + WRITE_SIZE summed over every launch / iterations"}
        json.dump(res, open(os.path.join(out_dir, f"bench_cfg3_{mode}_pmc.json"), "w"), indent=1)
        if mode == "streaming":
            ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                    "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"]
            vals, grid = counters(run(out_dir, "streaming_mfma", ["--pmc"] + ctrs, args))
            m = {}
            for kname, cs in vals.items():
                if "k_cross_mfma" not in kname:
                    continue
                e = m.setdefault(short(kname), {"grid_size": grid[kname]})
                for c, v in cs.items():
                    use = v[2:] if len(v) > 2 else v
                    e[c + "_mean"] = sum(use) / len(use)
                    e["launches"] = len(use)
                if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and "GRBM_GUI_ACTIVE_mean" in e:
                    # GRBM_GUI_ACTIVE sums 8 XCD instances; 1024 SIMDs (256 CUs x 4)
                    e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (e["GRBM_GUI_ACTIVE_mean"] / 8.0 * 1024.0)
            json.dump(m, open(os.path.join(out_dir, "bench_cfg3_streaming_mfma_pmc.json"), "w"), indent=1)
        # the bench line of the --stats run, for cross-checking launch_ms against the csv
        with open(os.path.join(out_dir, f"{mode}_stats.log")) as fh:
            lines = [ln for ln in fh if ln.startswith("{\"metric\"")]
        if lines:
            open(os.path.join(out_dir, f"bench_cfg3_{mode}_line.json"), "w").write(lines[-1])
    print("summaries in", out_dir)


if __name__ == "__main__":
    main()
