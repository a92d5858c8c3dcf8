#!/bin/bash
# usage: tools/pmc_run.sh TAG "COUNTER COUNTER ..." -- bench args ...     (one rocprofv3 --pmc pass; summary json under gpurun_out/r2/)
TAG=$1; CTRS=$2; shift 3
mkdir -p gpurun_out/r2
D=$PWD/gpurun_out/r2/pmc_$TAG
rm -rf $D
ROOT=$PWD
(cd /tmp && TMPDIR=/tmp timeout -k 10 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $D -- python3 $ROOT/bench.py "$@" > $D.log 2>&1)
python3 tools/summarise_pmc.py gpurun_out/r2/pmc_$TAG.json $D > /dev/null
python3 - <<PY
import json
d=json.load(open("gpurun_out/r2/pmc_$TAG.json"))
for k,v in d.items():
    if k.startswith("__amd") or "demc" not in k: continue
    print(k[:90], {c[:-5]:round(x) for c,x in v.items() if c.endswith("_mean")}, "launches", [x for c,x in v.items() if c.startswith("launches_")][:1])
PY
