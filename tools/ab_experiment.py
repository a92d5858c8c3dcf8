#!/usr/bin/env python3
"""A/B runs against the EXPERIMENTS build of the library (make EXPERIMENTS=1 OUT=../libdemc_hip_exp.so: the only build that
reads the DEMC_* environment switches):   python3 tools/ab_experiment.py VAR v1,v2,... -- <bench.py arguments>
prints ms_per_step and the dominant kernel's launch_ms for every value of VAR (an empty value = switch unset)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, values = sys.argv[1], sys.argv[2].split(",")
# VAR = LIB: the values are library file names under differentialevolutionmcmc.jl_amd/ (compile-time variants built with
# make EXTRA=-D... OUT=../<name>) instead of values of an environment switch
args = sys.argv[sys.argv.index("--") + 1:]
def code_for(lib):
    return ("import sys, runpy; sys.path.insert(0, %r); import demc_amd; "
            "demc_amd._ffi.LIB_PATH = %r; sys.argv = ['bench.py'] + %r; runpy.run_path(%r, run_name='__main__')"
            % (ROOT, os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", lib), args, os.path.join(ROOT, "bench.py")))


for v in values:
    env = dict(os.environ)
    code = code_for(v if var == "LIB" else "libdemc_hip_exp.so")
    env.pop(var, None)
    if v != "" and var != "LIB":
        env[var] = v
    for kv in os.environ.get("AB_ENV", "").split():  # extra switches for every run, e.g. AB_ENV="DEMC_LR_GS=1"
        k, _, val = kv.partition("=")
        env[k] = val
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    note = [ln for ln in out.stderr.splitlines() if "chunks chosen" in ln or "experiment" in ln][:1]
    if not line:
        print(var, "=", v, "FAILED", out.stderr[-500:])
        continue
    r = json.loads(line[-1])  # (the compact line is the last one)
    rf = r.get("roofline") or {}
    print(f"{var}={v or '(unset)':>8}  ms_per_step {r['ms_per_step']:.4f}  launch_ms {rf.get('launch_ms', float('nan')):.4f}  frac {rf.get('frac', float('nan')):.3f}  {note}", flush=True)
