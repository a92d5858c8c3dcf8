#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 600 python tools/k1_stamps.py > gpurun_out/r2/stamps_lean_cfg3.txt 2>&1; grep -E "softmax|plan written;|indices|per-dim|MvNormal|in-kernel|accept|kernel end" gpurun_out/r2/stamps_lean_cfg3.txt
timeout -k 10 600 python tools/k1_stamps.py --n-groups 32 --np 64 --dim 8 --nobs 10000 --mode streaming > gpurun_out/r2/stamps_lean_cfg2.txt 2>&1
tail -16 gpurun_out/r2/stamps_lean_cfg2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "lean or streaming_resident or plain_instance or resident" 2>&1 | tail -3
timeout -k 10 300 python bench.py --config cfg2 --steps 200 --warmup 50 --no-cpu-baseline --accuracy-iters 0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"device_ms_per_iter": [0-9.]*\|"frac": [0-9.]*' | head -4
timeout -k 10 300 python bench.py --mode suffstat --steps 200 --warmup 50 --no-cpu-baseline --accuracy-iters 0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"device_ms_per_iter": [0-9.]*\|"frac": [0-9.]*' | head -4
