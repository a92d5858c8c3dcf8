#!/usr/bin/env python3
"""Collects the rocprofv3 evidence behind bench.py's `roofline` objects and DESIGN.md section 6, on the GPU box:

    python3 tools/collect_profiles.py gpurun_out/profiles_rNN [row ...]     # summaries are also installed under profiles/rNN/

A row is `headline` (the default command) or a name of bench.ROWS (default: all of them); the profiled command is
`python3 bench.py <the row's flags> --rows none --no-cpu-baseline --accuracy-iters 0 --no-roofline` (the iterations once, without
HIP events: the un-profiled line at the end drops `--no-roofline`), i.e. what the driver's
`python3 bench.py --gpus 1` measures for that row.  For each row:
  * `rocprofv3 --kernel-trace --stats`                              -> bench_<row>_kernel_stats.csv
  * `--kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`  (separate passes, as MI355X_MICROARCH.md prescribes; FETCH_SIZE
    is doubled by the readers of the json: gfx950 counts half the bytes of wide streaming reads)   -> bench_<row>_pmc.json
  * once more with the pipe counters of the dominant kernel (matrix pipe for the MvNormal stream, VALU / LDS otherwise)
                                                                    -> bench_<row>_pipe_pmc.json
  * LAST an un-profiled run that reads the summaries just written   -> bench_<row>_line.json
Each program is started directly after `--` (no shell / env hop).  `dominant` in the pmc json names the kernel bench.py's
roofline is about and its HBM bytes per launch, or per ITERATION for the resident kernels (one launch covers a varying number
of iterations: all launches of the run together span warmup + steps iterations)."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dominant_pattern(over):
    """kernel-name pattern of the kernel the row's roofline is about (the default sampler on MvNormal-full runs in the lean
    resident kernel k_res_mvn; other samplers in k_propose<..., RES>)"""
    cfg, mode = over.get("config", "cfg3"), over.get("mode", "direct")  # (bench.py's default mode: the headline is DIRECT since round 6)
    if cfg == "cfg3" and mode == "streaming":
        return "k_cross_mfma"
    if cfg == "cfg3" and mode == "direct":
        return "k_direct_mvn"
    if cfg == "cfg2" and mode == "direct":
        return "k_res_mvn|k_propose<|k_direct_mvn"  # (a population this small: the residual loop inside the streaming-resident lean kernel)
    if cfg in ("cfg2", "cfg3"):
        return "k_res_mvn|k_propose<"  # (history partners: the lean body past burn-in, k_propose<256,false,...> inside it)
    return {"cfg4": "k_longrow|k_frozen_sweep", "cfg5": "k_lba_wave|k_lba_loglike|k_obs_loglike", "cfg1": "k_res_obs|k_propose<", "mvn30": "k_res_mvn|k_propose<"}[cfg]


MFMA_CTRS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE",
             "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"]
VALU_CTRS = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
             "GRBM_GUI_ACTIVE"]


def run(out_dir, tag, prof_args, bench_args):
    d = os.path.join(out_dir, tag)
    if os.environ.get("COLLECT_REUSE") == "1" and glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
        return d  # reduce the raw passes of an earlier collection again (no GPU needed)
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace"] + prof_args + ["--output-format", "csv", "-d", d, "--", "python3",
                                                        os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(out_dir, tag + ".log"), "w") as log:
        rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=600).returncode
    if rc != 0:  # (a profiled process that fails AFTER its outputs were generated is reported, not fatal)
        print(f"  pass {tag}: exit status {rc}", flush=True)
        if not glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
            raise SystemExit(f"pass {tag} produced no output")
    print("  pass", tag, "done", flush=True)
    return d


def counters(d):
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


def dominant_kernel(names, pat, resident):
    """the kernel of the run that matches the pattern: for k_propose, the resident / streaming-resident instance (its 4th
    template argument `true`) -- the initial evaluation runs in a non-resident instance and is not the roofline's subject"""
    for alt in pat.split("|"):
        cand = [n for n in names if alt in n]
        if cand:
            pat = alt
            break
    if pat == "k_propose<" and resident:
        res = [n for n in cand if len(n.split(",")) >= 4 and n.split(",")[3].strip().startswith("true")]
        cand = res or cand
    return max(cand, key=lambda n: names[n]) if cand else None


def main():
    sys.path.insert(0, ROOT)
    import bench
    rows = dict([("headline", dict(steps=20, warmup=5))] + bench.ROWS)
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0], formatter_class=argparse.RawDescriptionHelpFormatter,
                                 epilog="rows: " + " ".join(rows))
    ap.add_argument("out_dir", help="scratch directory for the raw passes (e.g. gpurun_out/profiles_r04)")
    ap.add_argument("rows", nargs="*", help="row names (default: all)")
    ap.add_argument("--no-pipe", action="store_true", help="skip the pipe-counter pass")
    opt = ap.parse_args()
    unknown = [r for r in opt.rows if r not in rows]
    if unknown:
        ap.error(f"unknown row(s) {unknown}; known: {' '.join(rows)}")
    fingerprint = bench.source_fingerprint()  # bench.py quotes a summary only for the sources it was collected on
    out_dir = os.path.abspath(opt.out_dir)
    os.makedirs(out_dir, exist_ok=True)
    for tag in (opt.rows or list(rows)):
        over = rows[tag]
        cfg, mode = over.get("config", "cfg3"), over.get("mode", "direct")
        n_iters = over.get("steps", 20) + over.get("warmup", 5)
        pattern = dominant_pattern(over)
        print(f"{tag}: {over}", flush=True)
        args = bench.row_flags(over) + ["--rows", "none", "--no-cpu-baseline", "--accuracy-iters", "0"]
        # the profiled passes run the iterations ONCE (--no-roofline: no HIP events, no instrumented repeat -- bench.two_pass);
        # the un-profiled line at the end is the plain command
        prof = args + ["--no-roofline"]
        d = run(out_dir, f"{tag}_stats", ["--stats"], prof)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(out_dir, f"bench_{tag}_kernel_stats.csv"))
        res, totals, fetch_totals = {}, defaultdict(float), defaultdict(float)
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            vals, grid = counters(run(out_dir, f"{tag}_{ctr}", ["--pmc", ctr], prof))
            for kname, cs in vals.items():
                v = cs[ctr]
                e = res.setdefault(short(kname), {"grid_size": grid[kname]})
                use = v[2:] if len(v) > 2 else v  # skip the first launches (initial evaluation, cold caches)
                e[ctr + "_KB_mean"] = sum(use) / len(use)
                e["launches_" + ctr] = len(use)
                e[ctr + "_KB_total"] = sum(v)
                e["launches_total"] = len(v)
                totals[short(kname)] += (2.0 if ctr == "FETCH_SIZE" else 1.0) * sum(v) * 1024.0
                if ctr == "FETCH_SIZE":
                    fetch_totals[short(kname)] += 2.0 * sum(v) * 1024.0
        dom = dominant_kernel(totals, pattern, resident=True)
        if dom:
            e = res[dom]
            resident = ("k_propose<" in pattern or "k_res_obs" in pattern) and over.get("partners") != "history"
            rec = {"kernel": dom, "note": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 half-count of wide streaming reads)"}
            if resident:  # a launch covers several iterations: reduce to bytes per iteration over the whole run
                rec["bytes_per_iteration"] = totals[dom] / n_iters
                rec["iterations"] = n_iters
            else:
                rec["bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KB_mean"] + e["WRITE_SIZE_KB_mean"]) * 1024.0
                rec["fetch_bytes_per_launch"] = 2.0 * e["FETCH_SIZE_KB_mean"] * 1024.0
            rec["launches"] = e["launches_total"]
            also = [n for n in totals if "k_frozen_sweep" in n and n != dom]
            if cfg == "cfg4" and also:
                # a blocked long-row run alternates two kernels (k_frozen_sweep<256> for the hyper-parameter sweep, k_frozen_sweep<256,big>
                # or k_longrow for the subject sweep): the row's traffic is both kernels' -- reduced to bytes per iteration over the
                # whole run, like the resident rows
                rec.pop("bytes_per_launch", None)
                rec.pop("fetch_bytes_per_launch", None)
                rec["bytes_per_iteration"] = (totals[dom] + sum(totals[n] for n in also)) / n_iters
                rec["fetch_bytes_per_iteration"] = (fetch_totals[dom] + sum(fetch_totals[n] for n in also)) / n_iters
                rec["iterations"] = n_iters
                rec["also"] = also
                rec["launches"] = e["launches_total"] + sum(res[n]["launches_total"] for n in also)
            res["dominant"] = rec
        res["source_sha16"] = fingerprint
        res["command"] = "python3 bench.py " + " ".join(prof)
        json.dump(res, open(os.path.join(out_dir, f"bench_{tag}_pmc.json"), "w"), indent=1)
        if not opt.no_pipe:
            # pipe counters of the dominant kernel
            mfma = (cfg in ("cfg2", "cfg3")) and mode == "streaming"  # (DIRECT is vector-pipe work: VALU counters)
            ctrs = MFMA_CTRS if mfma else VALU_CTRS
            vals, grid = counters(run(out_dir, f"{tag}_pipe", ["--pmc"] + ctrs, prof))
            m = {}
            for kname, cs in vals.items():
                if "demc" not in kname:
                    continue
                e = m.setdefault(short(kname), {"grid_size": grid[kname]})
                for c, v in cs.items():
                    use = v[2:] if len(v) > 2 else v
                    e[c + "_mean"] = sum(use) / len(use)
                    e["launches"] = len(use)
                if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and e.get("GRBM_GUI_ACTIVE_mean", 0) > 0:
                    # GRBM_GUI_ACTIVE sums 8 XCD instances; 1024 SIMDs (256 CUs x 4)
                    e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (e["GRBM_GUI_ACTIVE_mean"] / 8.0 * 1024.0)
                if "SQ_ACTIVE_INST_VALU_mean" in e and e.get("GRBM_GUI_ACTIVE_mean", 0) > 0:
                    # SQ_ACTIVE_INST_VALU: cycles (x4) a SIMD spent issuing VALU work, summed over the chip
                    e["valu_busy_frac"] = 4.0 * e["SQ_ACTIVE_INST_VALU_mean"] / (e["GRBM_GUI_ACTIVE_mean"] / 8.0 * 1024.0)
                if "SQ_INSTS_VALU_mean" in e and e.get("SQ_WAVES_mean", 0) > 0:
                    e["valu_insts_per_wave"] = e["SQ_INSTS_VALU_mean"] / e["SQ_WAVES_mean"]
                if "SQ_LDS_IDX_ACTIVE_mean" in e and e.get("GRBM_GUI_ACTIVE_mean", 0) > 0:
                    # cycles the LDS pipe of a CU was working, summed over the 256 CUs
                    e["lds_busy_frac"] = e["SQ_LDS_IDX_ACTIVE_mean"] / (e["GRBM_GUI_ACTIVE_mean"] / 8.0 * 256.0)
            m["source_sha16"] = fingerprint
            json.dump(m, open(os.path.join(out_dir, f"bench_{tag}_pipe_pmc.json"), "w"), indent=1)
        # The bench line LAST, from an un-profiled run that reads the summaries just written: the summaries are installed
        # under profiles/<round>/ of this tree first (bench.PROFILE_ROUND), then bench.py runs once more.
        inst = os.path.join(ROOT, "profiles", bench.PROFILE_ROUND)
        os.makedirs(inst, exist_ok=True)
        for suffix in ("pmc.json", "pipe_pmc.json", "kernel_stats.csv"):
            src = os.path.join(out_dir, f"bench_{tag}_{suffix}")
            if os.path.exists(src):
                shutil.copy(src, inst)
        with open(os.path.join(out_dir, f"{tag}_line.log"), "w") as log:
            out = subprocess.run(["python3", os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=log, text=True, timeout=600)
        # (bench.py prints {"detail": <everything>} first and the compact line last: the committed record is the full one)
        lines = [json.dumps(json.loads(ln)["detail"]) for ln in out.stdout.splitlines() if ln.startswith("{\"detail\"")]
        if lines:
            open(os.path.join(out_dir, f"bench_{tag}_line.json"), "w").write(lines[-1] + "\n")
            shutil.copy(os.path.join(out_dir, f"bench_{tag}_line.json"), inst)
    print("summaries in", out_dir)


if __name__ == "__main__":
    main()
