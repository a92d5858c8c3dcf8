#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
step() { local name=$1 to=$2; shift 2; echo "== $name" | tee -a gpurun_out/r2/session.log
  timeout -k 10 $to "$@" > gpurun_out/r2/$name.log 2>&1; local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/r2/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout -> stop"; exit 1; fi; return 0; }
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; cat /proc/loadavg
step t_stream 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "streaming_resident or default_sampler or large_tile or suffstat or fused"
tail -15 gpurun_out/r2/t_stream.log
step bench_cfg2b 300 python bench.py --config cfg2 --steps 200 --warmup 50 --no-cpu-baseline
step bench_cfg2c 300 python bench.py --config cfg2 --steps 200 --warmup 50 --no-cpu-baseline --no-roofline
grep -h '^{"metric"' gpurun_out/r2/bench_cfg2b.log gpurun_out/r2/bench_cfg2c.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline'] or {}
    print(r['config']['workload'][:50], '| value %.3e ms/step %.4f'%(r['value'], r['ms_per_step']), rf.get('frac'), rf.get('per_kernel_ms_per_iter'), rf.get('launches_by_class'), r['accuracy'].get('posterior_mean_l1_rel'))
"
