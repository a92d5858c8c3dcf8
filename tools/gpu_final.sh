#!/bin/bash
mkdir -p gpurun_out/r2
run() { local name=$1; shift; timeout -k 10 420 python bench.py "$@" > gpurun_out/r2/final_$name.log 2>&1; echo "$name rc=$?"; }
run cfg3
run cfg3_driver --steps 20 --warmup 5
run cfg3_suffstat --mode suffstat
run cfg2 --config cfg2
run cfg4 --config cfg4 --steps 100 --warmup 20
run cfg4_full --config cfg4 --n-groups 128 --steps 40 --warmup 10 --no-cpu-baseline
run cfg5 --config cfg5 --steps 30 --warmup 5
grep -h '^{"metric"' gpurun_out/r2/final_*.log | python -c "
import sys,json
for ln in sys.stdin:
    r=json.loads(ln); rf=r['roofline'] or {}; cb=r['cpu_baseline'] or {}
    print(r['config']['workload'][:64], '| steps', r['steps'], '| value %.4e ms/step %.4f'%(r['value'], r['ms_per_step']), 'frac %.3f'%rf.get('frac',0), rf.get('bound'), 'dev_ms %.4f'%rf.get('device_ms_per_iter',0), 'traffic', rf.get('traffic'), 'waste', rf.get('wasted_traffic_ratio'), '| acc', r['accuracy'].get('posterior_mean_l1_rel'), '| cpu %.3e (%s thr) 1thr %.3e'%(cb.get('value',0), cb.get('cores'), cb.get('value_single_thread',0)))
"
