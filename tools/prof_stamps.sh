#!/bin/bash
# kernel durations (rocprofv3) of the diagnostic build's run, next to its in-kernel stamps: calibrates the stamp clock
out=$PWD/gpurun_out/r2b/stamps_trace
rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/k1_stamps.py --config cfg4 > $out/run.log 2>&1
tail -8 $out/run.log
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "longrow" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; last 8 durations (us):", [round(x, 1) for x in d[-8:]])
PY
