#!/usr/bin/env python3
"""Quick GPU check of the wide kernels (K > 112 or d > 64: k_assign_wide / k_rtz_wide / k_ridge_apply_wide) against the
oracle on a small case of BASELINE configs[4]'s shape: `python scripts/gpu_wide_check.py [cells]` (default 6000)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HMX_UPDATE_ORDER"] = "device"


class _Env:
    def setenv(self, k, v):
        os.environ[k] = v


if __name__ == "__main__":
    from test_parity_gpu import _bench_path_case
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    _bench_path_case(n, 200, 32, 200, _Env(), ridge_dtype=np.float64, rounds=(2, 2))
    print("wide check ok")
