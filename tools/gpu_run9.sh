#!/bin/bash
mkdir -p gpurun_out/r2/stats_ss
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r2/stats_ss -- python3 $OLDPWD/bench.py --mode suffstat --steps 200 --warmup 50 --no-cpu-baseline --accuracy-iters 0 > $OLDPWD/gpurun_out/r2/stats_ss.log 2>&1)
cat gpurun_out/r2/stats_ss/*/*kernel_stats.csv | cut -c1-160 | head -8
grep -o '"ms_per_step": [0-9.]*\|"device_ms_per_iter": [0-9.]*\|"launches_by_class": {[^}]*}' gpurun_out/r2/stats_ss.log
