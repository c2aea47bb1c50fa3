#!/bin/bash
# cycle stamps of k_round, k_rtz3 (C3), k_rtzw2 and -- with HMX_WIDE_SWEEP=1 -- k_round_wide (configs[4] shard)
# (profiling build: python -m harmonypy_amd._build -o build/libhmx_prof.so -DHMX_ROUND_PROF -DHMX_RTZ3_PROF)
HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
HMX_WIDE_SWEEP=1 HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "k_round_wide prof\]"
