#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 tests/test_gpu_edge_cases.py -k "not nothing" > gpurun_out/r2/t_stream.log 2>&1 || { tail -30 gpurun_out/r2/t_stream.log; exit 1; }
tail -3 gpurun_out/r2/t_stream.log
timeout -k 10 600 true
true
timeout -k 10 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2/bench_cfg4b.log 2>&1
grep -h '^{"metric"' gpurun_out/r2/bench_cfg4b.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline'] or {}
    print(r['config']['workload'][:50], '| value %.3e ms/step %.4f'%(r['value'], r['ms_per_step']), rf.get('frac'), rf.get('per_kernel_ms_per_iter'))
"
