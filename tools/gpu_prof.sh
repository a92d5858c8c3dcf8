#!/bin/bash
mkdir -p gpurun_out/profiles_r02
timeout -k 10 1150 python3 tools/collect_profiles.py gpurun_out/profiles_r02 cfg2:streaming cfg4:streaming cfg5:streaming > gpurun_out/profiles_r02/collect.log 2>&1
tail -30 gpurun_out/profiles_r02/collect.log
ls gpurun_out/profiles_r02 | head -60
