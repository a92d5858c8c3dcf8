#!/usr/bin/env python3
"""The ONE current measurement table of DESIGN.md section 6, generated from the committed records of a round:
    python3 tools/design_tables.py profiles/r05 > /tmp/table.md          (or: ... profiles/r05 --into DESIGN.md, which replaces the table in place)
Per row of bench.ROWS (and the headline): value, ms per step, the dominant kernel's launch time and roofline fraction from
`bench_<row>_line.json` (the un-profiled run's full record), the rocprofv3 --stats average of the same kernel from
`bench_<row>_kernel_stats.csv`, and the counters (`bench_<row>_pmc.json`, `bench_<row>_pipe_pmc.json`).  Nothing is typed by
hand into that table: a number that is not in profiles/<round>/ is not in DESIGN.md."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def g(x, n=4):
    return "—" if x is None else f"{x:.{n}g}"


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("demc::", "")


def main(d):
    print("| row | particle-updates/s | ms / step | kernels (`demc_last_kernels`) | per launch (HIP events; `--stats` average) | `roofline.frac` (bound) | counters |")
    print("|---|---|---|---|---|---|---|")
    for row in ["headline"] + [n for n, _ in bench.ROWS]:
        ln = os.path.join(d, f"bench_{row}_line.json")
        if not os.path.exists(ln):
            continue
        rec = json.load(open(ln))
        if row != "headline" and rec.get("rows"):
            rec = rec["rows"][0]
        rf = rec.get("roofline") or {}
        stats = ""
        st = os.path.join(d, f"bench_{row}_kernel_stats.csv")
        if os.path.exists(st):
            rows = sorted((r for r in csv.DictReader(open(st)) if "demc" in r["Name"]), key=lambda r: -float(r["TotalDurationNs"]))
            if rows:
                r = rows[0]
                stats = f"; `{short(r['Name'])[:40]}` {r['Calls']} calls, {g(float(r['AverageNs']) / 1e3)} µs"
        ctr = []
        if rf.get("frac_survey") is not None:  # STREAMING rows: the executed MFMA flop next to SURVEY 8d's unit (above 1: not the survey's work)
            ctr.append(f"`frac_executed` {g(rf.get('frac_executed'), 3)}, `frac_survey` {g(rf['frac_survey'], 3)}")
        if rf.get("shader_clock_mhz") is not None:  # DIRECT rows: the clock the vector pipe held (in-kernel) and the fraction at that clock
            ctr.append(f"shader clock {g(rf['shader_clock_mhz'])} MHz in-kernel: {g(rf.get('frac_at_clock'), 3)} of the peak at that clock" +
                       (f", {g(rf['frac_of_mix_ceiling_at_clock'], 3)} of the add+fma mix's ceiling" if rf.get("frac_of_mix_ceiling_at_clock") is not None else ""))
        if rf.get("traffic") is not None:
            ctr.append(f"{g(rf['traffic'] / 1e6)} MB/launch ({g(rf.get('wasted_traffic_ratio'), 3)}× algorithmic)")
        if rf.get("counter_frac") is not None:
            ctr.append(f"{g(rf['counter_frac'], 3)} of HBM by counter bytes")
        if rf.get("fetch_bytes_per_update") is not None:
            ctr.append(f"FETCH {g(rf['fetch_bytes_per_update'])} B/update (gather {g(rf.get('gather_bytes_per_update'))})")
        pp, pm = os.path.join(d, f"bench_{row}_pipe_pmc.json"), os.path.join(d, f"bench_{row}_pmc.json")
        if os.path.exists(pp):
            # the pipe counters of the row's DOMINANT kernel (named by the traffic passes; else the kernel with the most launches x cycles)
            dom = (json.load(open(pm)).get("dominant") or {}).get("kernel") if os.path.exists(pm) else None
            pipes = {k: v for k, v in json.load(open(pp)).items() if isinstance(v, dict)}
            best = pipes.get(dom) or pipes.get("demc::" + str(dom))
            if best is None and pipes:
                best = max(pipes.values(), key=lambda v: v.get("SQ_WAVE_CYCLES_mean", 0.0) * v.get("launches", 1))
            if best:
                for key, label in (("mfma_busy_frac", "MFMA busy"), ("valu_busy_frac", "VALU busy"), ("lds_busy_frac", "LDS busy")):
                    if best.get(key) is not None:
                        ctr.append(f"{label} {g(best[key], 3)}")
        kern = rec.get("kernels") or (rf.get("kernel") or "")[:60]
        print(f"| `{row}` | {g(rec.get('value'))} | {g(rec.get('ms_per_step'))} | `{kern}` | {g((rf.get('launch_ms') or 0) * 1e3)} µs{stats} | "
              f"**{g(rf.get('frac'), 3)}** ({rf.get('bound')}) | {'; '.join(ctr) or '—'} |")


def install(d, path):
    """replace the table in `path` (from its header line to the last row) with the one generated from d"""
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        main(d)
    new = buf.getvalue().rstrip("\n").split("\n")
    lines = open(path).read().split("\n")
    a = next(i for i, ln in enumerate(lines) if ln.startswith("| row | particle-updates/s"))
    b = a
    while b < len(lines) and lines[b].startswith("|"):
        b += 1
    open(path, "w").write("\n".join(lines[:a] + new + lines[b:]))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--into":  # python3 tools/design_tables.py profiles/r05 --into DESIGN.md
        install(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1])
