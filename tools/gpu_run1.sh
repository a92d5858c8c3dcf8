#!/bin/bash
# GPU session: tests, then bench lines.  A step that times out (124/137) ends the session.
set -o pipefail
mkdir -p gpurun_out/r2
step() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/r2/session.log
  timeout -k 10 $to "$@" > gpurun_out/r2/$name.log 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/r2/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout -> stop" | tee -a gpurun_out/r2/session.log; exit 1; fi
  return 0
}
step tests 1000 python -m pytest tests -m gpu -q -x --timeout 600
tail -5 gpurun_out/r2/tests.log
step bench_cfg3 300 python bench.py --steps 40 --warmup 10
step bench_cfg3_suff 300 python bench.py --steps 200 --warmup 50 --mode suffstat --no-cpu-baseline
step bench_cfg2 300 python bench.py --config cfg2 --steps 200 --warmup 50
step bench_cfg4 300 python bench.py --config cfg4 --steps 50 --warmup 10
step bench_cfg5 300 python bench.py --config cfg5 --steps 20 --warmup 5
grep -h '^{"metric"' gpurun_out/r2/bench_*.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline']
    print(r['config']['workload'][:60], '| value %.3e ms/step %.4f frac %.3f (%s) cpu %.3e'%(r['value'], r['ms_per_step'], rf['frac'], rf['bound'], (r['cpu_baseline'] or {}).get('value',0)))
"
