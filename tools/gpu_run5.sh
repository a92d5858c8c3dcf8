#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_api.py -q -x --timeout 600 -k "lognormal or run_lba or full_share" > gpurun_out/r2/t_gates.log 2>&1
tail -40 gpurun_out/r2/t_gates.log
