#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2/tests_full.log 2>&1
tail -25 gpurun_out/r2/tests_full.log
