#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "streaming_resident or default_sampler or large_tile or suffstat or fused or resident" > gpurun_out/r2/t_stream.log 2>&1 || { tail -30 gpurun_out/r2/t_stream.log; exit 1; }
tail -3 gpurun_out/r2/t_stream.log
timeout -k 10 600 python tools/k1_stamps.py --n-groups 32 --np 64 --dim 8 --nobs 10000 --mode streaming > gpurun_out/r2/stamps_cfg2_stream.txt 2>&1
cat gpurun_out/r2/stamps_cfg2_stream.txt | tail -18
timeout -k 10 300 python bench.py --config cfg2 --steps 200 --warmup 50 > gpurun_out/r2/bench_cfg2b.log 2>&1
grep -h '^{"metric"' gpurun_out/r2/bench_cfg2b.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline'] or {}
    print(r['config']['workload'][:50], '| value %.3e ms/step %.4f'%(r['value'], r['ms_per_step']), rf.get('frac'), rf.get('per_kernel_ms_per_iter'), r['cpu_baseline']['value'], r['cpu_baseline']['cores'], r['cpu_baseline']['value_single_thread'])
"
