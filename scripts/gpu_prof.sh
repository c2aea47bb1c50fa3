#!/bin/bash
# cycle stamps of k_round and of the streaming R^T.Z kernels (profiling build: python -m harmonypy_amd._build -o build/libhmx_prof.so -DHMX_ROUND_PROF -DHMX_RTZ3_PROF)
HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
HMX_LIB=$PWD/build/libhmx_prof.so timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-convergence --no-lisi 2>&1 | grep "prof\]"
