#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2/tests_full.log 2>&1
tail -6 gpurun_out/r2/tests_full.log
for c in cfg2; do
timeout -k 10 300 python bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --accuracy-iters 0 > gpurun_out/r2/bench_${c}c.log 2>&1
done
timeout -k 10 300 python bench.py --mode suffstat --steps 200 --warmup 50 --no-cpu-baseline --accuracy-iters 0 > gpurun_out/r2/bench_cfg3sc.log 2>&1
grep -h '^{"metric"' gpurun_out/r2/bench_cfg4c.log gpurun_out/r2/bench_cfg2c.log gpurun_out/r2/bench_cfg3sc.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline'] or {}
    print(r['config']['workload'][:50], '| value %.3e ms/step %.4f'%(r['value'], r['ms_per_step']), rf.get('frac'), rf.get('device_ms_per_iter'))
"
