#!/bin/bash
mkdir -p gpurun_out/profiles_r02 gpurun_out/r2
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r2/tests_full.log 2>&1; tail -3 gpurun_out/r2/tests_full.log
timeout -k 10 1100 python3 tools/collect_profiles.py gpurun_out/profiles_r02 cfg3:suffstat cfg2:streaming cfg4:streaming > gpurun_out/profiles_r02/collect.log 2>&1
tail -5 gpurun_out/profiles_r02/collect.log
timeout -k 10 300 python tools/k1_stamps.py > gpurun_out/r2/stamps_cfg3_suff.txt 2>&1
timeout -k 10 300 python tools/k1_stamps.py --n-groups 32 --np 64 --dim 8 --nobs 10000 --mode streaming > gpurun_out/r2/stamps_cfg2_stream.txt 2>&1
timeout -k 10 300 python tools/k1_stamps.py --config cfg4 > gpurun_out/r2/stamps_cfg4.txt 2>&1
bash tools/gpu_final.sh
