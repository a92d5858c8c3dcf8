#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -q --timeout 300 -k "asynchronous or rccl" > gpurun_out/r2/t_async.log 2>&1
grep -E "passed|failed|FAILED|array [0-9]" gpurun_out/r2/t_async.log | head -20
