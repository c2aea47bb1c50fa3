#!/usr/bin/env python3
"""CPU arm of the bench line, re-measured in the build container.

Runs `bench.py --cpu-only` twice per sample size -- harmonypy itself (kind "reference", HMX_REFERENCE_PATH) and the
NumPy oracle (kind "port") -- on the same synthetic C3-shaped sample, the same centroids and the same thread count, each
in a fresh process, and writes profiles/r06_cpu_baseline_calibration.json.  A GPU box has no harmonypy: bench.py times
the port there and multiplies by `reference_over_port` from this file, labelled as a cross-host extrapolation.

    python scripts/cpu_calibration.py [--sizes 200000,1000000] [--out profiles/r06_cpu_baseline_calibration.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(size, reference):
    env = dict(os.environ)
    env.pop("HMX_REFERENCE_PATH", None)
    if reference:
        env["HMX_REFERENCE_PATH"] = "/root/reference"

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-only", "--cpu-sample", str(size)],
                         env=env, check=True, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="200000,1000000")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_cpu_baseline_calibration.json"))
    args = ap.parse_args()
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    doc = {"what": "bench.py --cpu-only in the build container: harmonypy v0.2.0 itself (kind \"reference\", "
                   "HMX_REFERENCE_PATH=/root/reference, device='cpu') and the NumPy oracle (kind \"port\") on the same synthetic "
                   "C3-shaped sample (50 PCs, 8 batches, K=100), same centroids, 10 rounds + ridge, same thread count, "
                   "one fresh process each",
           "host_cpus": os.cpu_count(), "commit": head, "samples": {}}
    for size in [int(s) for s in args.sizes.split(",")]:
        ref = run(size, True)
        port = run(size, False)
        doc["samples"][str(size)] = {"reference": ref, "port": port, "reference_over_port": ref["value"] / port["value"]}
        print(size, "reference", round(ref["value"]), "port", round(port["value"]), file=sys.stderr)
        with open(args.out, "w") as f:   # after every size: a long run that is cut short still leaves what it measured
            doc["reference_over_port"] = doc["samples"][str(min(int(s) for s in doc["samples"]))]["reference_over_port"]
            json.dump(doc, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
