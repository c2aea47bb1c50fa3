#!/usr/bin/env python3
"""bench.py -- throughput of the Harmony iteration engine on MI355X.

    python bench.py --gpus 1 --steps K --warmup W            (defaults finish in ~2-3 minutes)
    python bench.py --gpus N ...                             (launches its own N ranks, one per GPU, over RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (the driver's form)

Metric (BASELINE.json): cells/sec/Harmony-iteration = cells x iterations / T_harmonize, inputs
resident in HBM when the timed region starts.  A *step* is one Harmony iteration
(harmony.py:421-432): `--rounds` k-means rounds of harmony.py:443-453 (fixed, default 10 = the
mean of the reference's C3 schedule [20,10,5,5]; the objective is read back every round exactly
as the reference's `.item()` calls do, the convergence test is evaluated but not acted on) plus
one moe_correct_ridge (harmony.py:535-569).  The per-round update order is produced on the
device inside the timed region.

Workload at --gpus 1: BASELINE.json configs[2] (C3, the roofline point): synthetic 1M cells x
50 PCs, 8 batches, K=100.  `--config c2` selects configs[1] (69k x 50, 4 batches, K=30).
With N > 1 ranks the default is the job BASELINE.json names for several GPUs: configs[3], 10M cells x 50 PCs,
16 batches, K=100, cells sharded evenly over the N ranks (STRONG scaling: 10M / N cells per GPU; its one-GPU point
is the `configs_3_on_one_gpu` entry of the default line); `--config c5` shards configs[4] (10M x 200 PCs, K=200) the
same way, `--config c3` keeps 1M cells per GPU (weak scaling).  A run that cannot bring up N ranks on N devices
exits non-zero instead of reporting fewer GPUs.

One JSON line on stdout (rank 0).  Besides the contract fields:
  roofline     -- the dominant kernel (k_round, one persistent launch per update_R sweep):
                  algorithmic bytes per launch (cells x (4d + 4K + 4)) / average
                  launch time from HIP events on the engine's stream, against 8 TB/s HBM.
  cpu_baseline -- the NumPy oracle (a port of the reference's torch-CPU path) timed on this
                  box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (cells per GPU, PCs, batches, clusters)
    "c2": (69_000, 50, 4, 30),
    "c3": (1_000_000, 50, 8, 100),
    # per-GPU shards of the 8-GPU configurations (BASELINE configs[3], configs[4])
    "c4": (1_250_000, 50, 16, 100),
    "c5": (1_250_000, 200, 32, 200),
    # all 10M cells of BASELINE configs[3] on ONE GPU (R = 4.5 GB of the 288 GB)
    "c4x1": (10_000_000, 50, 16, 100),
    # ... and half of them: the per-GPU share of configs[3] on 2 GPUs (blocks of 250k cells: still larger than the sweep's grid)
    "c4x2": (5_000_000, 50, 16, 100),
}
CUSTOM_SHAPES = set()
if os.environ.get("BENCH_SHAPE"):   # experiments only: "name=N,d,B,K" replaces a configuration's shape (the line then says so: no BASELINE label)
    _n, _v = os.environ["BENCH_SHAPE"].split("=")
    CONFIGS[_n] = tuple(int(x) for x in _v.split(","))
    CUSTOM_SHAPES.add(_n)


def config_label(name):
    """`BASELINE configs[i] (NAME)`, or an explicit note when BENCH_SHAPE replaced the configuration's shape."""
    if name in CUSTOM_SHAPES:
        return f"EXPERIMENT SHAPE (BENCH_SHAPE, not a BASELINE configuration; slot {name.upper()})"
    return f"BASELINE configs[{CONFIG_INDEX[name]}] ({name.upper()})"
CONFIG_INDEX = {"c2": 1, "c3": 2, "c4": 3, "c5": 4, "c4x1": 3, "c4x2": 3}
# cells of the whole job of the configurations BASELINE.json defines over several GPUs (sharded evenly: strong scaling)
JOB_CELLS = {"c4": 10_000_000, "c5": 10_000_000}


def self_launch(args_list, n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand their exit status on.  Returns the command (for tests)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)


def resolve_workload(config, gpus):
    """(config name, cells of this job per GPU, scaling) for `--config` (None = default) on `gpus` ranks."""
    if config is None:
        config = "c3" if gpus == 1 else "c4"
    if gpus > 1 and config in JOB_CELLS:
        return config, JOB_CELLS[config] // gpus, "strong"
    return config, CONFIGS[config][0], "weak"

TIMED_LAUNCH_STRIDE = 4   # the dominant kernel's launches bracketed with HIP events inside the timed region: every 4th (every 7th left 8 samples per default run: one delayed launch moved the average by 5 %)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense (= the f32 vector rate)
BF16X3_PEAK_TF = 2500.0 / 6.0   # fp32 products as six bf16 MFMAs on three exact bf16 terms per operand: the dense bf16 peak (~2.5 PFLOP/s) / 6
REFERENCE_AT_1M_SURVEY = 2.56e4   # harmonypy itself (device='cpu', 8 threads) at configs[2]'s size, cells/s/iteration


def synthetic_dataset(N, d, B, K, seed=0, cell_seed=None):
    """Seeded PC-like matrix with batch offsets (SURVEY.md §8d / BASELINE.md §3):
    PC scale 1/sqrt(1+j); T = max(K//2, 5) cell-type centres N(0, 3^2)*scale; batch offsets
    N(0,1)*scale; batch proportions Dirichlet(5); unit noise*scale; float32; one categorical
    column ``batch`` with labels b0..b{B-1}.  ``cell_seed`` (default: none) draws different cells
    from the same population: the shards of one sharded job."""
    rng = np.random.default_rng(seed)
    T = max(K // 2, 5)
    scale = (1.0 / np.sqrt(1.0 + np.arange(d))).astype(np.float32)
    centres = (rng.normal(0, 3.0, (T, d)) * scale).astype(np.float32)
    offsets = (rng.normal(0, 1.0, (B, d)) * scale).astype(np.float32)
    p = rng.dirichlet(5.0 * np.ones(B))
    if cell_seed is not None:
        rng = np.random.default_rng([seed, 1 + int(cell_seed)])
    batch = rng.choice(B, size=N, p=p).astype(np.int32)
    typ = rng.integers(0, T, size=N)
    Z = rng.standard_normal((N, d), dtype=np.float32)
    Z *= scale
    Z += centres[typ]
    Z += offsets[batch]
    labels = np.array([f"b{i}" for i in range(B)])
    meta = pd.DataFrame({"batch": pd.Categorical.from_codes(batch, categories=labels)})
    return Z, meta


def quick_centroids(Z, K, seed=0, sample=50_000):
    """k-means++ / Lloyd on a subsample (untimed initialisation; harmony.py:370-372 on a sample)."""
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(seed)
    idx = rng.choice(Z.shape[0], size=min(sample, Z.shape[0]), replace=False)
    Zs = Z[idx]
    Zs = Zs / np.linalg.norm(Zs, axis=1, keepdims=True)
    km = KMeans(n_clusters=K, init="k-means++", n_init=1, max_iter=25, random_state=seed).fit(Zs)
    return np.asarray(km.cluster_centers_.T, dtype=np.float32)  # d x K


def _reference_baseline(ref_path, Z, meta, K, rounds, Y0):
    """One Harmony iteration (`rounds` k-means rounds + ridge) of the reference itself, device='cpu', timed around its
    own harmonize() (harmony.py:419-435); the sklearn fit is replaced by the prepared centroids."""
    import logging
    sys.path.insert(0, ref_path)
    import harmonypy as hm
    import harmonypy.harmony as hh
    import torch
    logging.getLogger("harmonypy").setLevel(logging.WARNING)

    class FixedKMeans:
        def __init__(self, *a, **k):
            pass

        def fit(self, X):
            self.cluster_centers_ = Y0.T.astype(np.float64)
            return self
    timing = {}
    orig = hh.Harmony.harmonize

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        timing["dt"] = time.perf_counter() - t0
        return r
    hh.KMeans, hh.Harmony.harmonize = FixedKMeans, timed
    try:
        hm.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=1, max_iter_kmeans=rounds, epsilon_cluster=0.0,
                       epsilon_harmony=-1e30, verbose=False, random_state=0, device="cpu")
    finally:
        hh.Harmony.harmonize = orig
    return timing["dt"], int(torch.get_num_threads())


def cpu_baseline(d, B, K, rounds, sample_cells, seed=1):
    """The reference's CPU path on a bounded sample of the workload, one Harmony iteration.

    kind "reference": harmonypy itself (`run_harmony(..., device='cpu')`, harmony.py:49) when the environment variable
    HMX_REFERENCE_PATH names a checkout of it (the build container: /root/reference; a GPU box has none).
    kind "port": the NumPy oracle, a restatement of the same torch-CPU arithmetic, on min(32, cpus) BLAS threads; its
    rate relative to the reference's on the same sample and cores is kept in profiles/r06_cpu_baseline_calibration.json
    (scripts/cpu_calibration.py, measured in the build container, where both run: 200k and 1M cells)."""
    Z, meta = synthetic_dataset(sample_cells, d, B, K, seed=seed)
    Y0 = quick_centroids(Z, K, seed=seed)
    what = f"{sample_cells} cells x {d} PCs, {B} batches, K={K}: 1 iteration = {rounds} rounds + ridge"
    ref_path = os.environ.get("HMX_REFERENCE_PATH")
    if not (ref_path and os.path.isdir(os.path.join(ref_path, "harmonypy"))):
        ref_path = None
        try:   # an installed harmonypy (site-packages) serves as well as a checkout
            import importlib.util
            spec = importlib.util.find_spec("harmonypy")
            if spec is not None and spec.origin and "harmonypy_amd" not in spec.origin:
                ref_path = os.path.dirname(os.path.dirname(spec.origin))
        except Exception:
            ref_path = None
    if ref_path:
        dt, threads = _reference_baseline(ref_path, Z, meta, K, rounds, Y0)
        return {"value": sample_cells / dt, "unit": "cells/sec/Harmony-iteration", "cores": threads, "kind": "reference",
                "sample": f"{what} in {dt:.1f} s (harmonypy at {ref_path}, device='cpu', torch threads={threads}, "
                          f"host has {os.cpu_count()} cpus)"}
    from oracle.harmony_oracle import OracleHarmony, prepare_inputs
    cal = None
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_baseline_calibration.json")))
    except Exception:
        pass
    # the thread count the reference/port ratio was measured at (8), so that the ratio applies to this very run
    cal_threads = int(cal["samples"]["200000"]["port"]["cores"]) if cal else 8
    threads = min(cal_threads, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=threads)
    except Exception:
        limit = None
    p = prepare_inputs(Z, meta, ["batch"], nclust=K)
    rng = np.random.default_rng(seed)
    oo = OracleHarmony(p["Z"], p["phi"], p["Pr_b"], p["sigma"], p["theta"], p["lamb"], K=K, run=False,
                       perm_source=lambda n: rng.permutation(n), forced_rounds=[rounds])
    oo.init_cluster(0, Y0)
    t0 = time.perf_counter()
    oo.cluster()
    oo.moe_correct_ridge()
    oo.check_convergence(1)
    dt = time.perf_counter() - t0
    if limit is not None:
        limit.restore_original_limits()
    out = {"value": sample_cells / dt, "unit": "cells/sec/Harmony-iteration", "cores": int(threads), "kind": "port",
           "sample": f"{what} in {dt:.1f} s (NumPy oracle: no harmonypy on this box (HMX_REFERENCE_PATH, site-packages), "
                     f"BLAS threads={threads}, host has {os.cpu_count()} cpus)"}
    if cal:
        out["reference_over_port"] = cal.get("reference_over_port")
        out["value_reference_equivalent"] = out["value"] * cal.get("reference_over_port", 1.0)
        out["value_reference_equivalent_is"] = ("a CROSS-HOST EXTRAPOLATION: this host's port rate x the reference/port ratio measured on the "
                                                "build container (8 cpus), where harmonypy itself runs -- not a measurement of the reference on this box")
        out["calibration"] = ("profiles/r06_cpu_baseline_calibration.json (scripts/cpu_calibration.py: harmonypy itself vs this port, same sample, "
                              "same 8 threads, 200k and 1M cells, build container)")
        big = cal.get("samples", {}).get("1000000")
        if big:   # the reference itself at the HEADLINE size, measured this round where it runs
            out["reference_at_1M_build_container"] = {"value": big["reference"]["value"], "unit": "cells/sec/Harmony-iteration",
                                                      "cores": big["reference"]["cores"], "port_same_host": big["port"]["value"],
                                                      "sample": big["reference"]["sample"]}
    return out


def newest_pmc(engine_version):
    """HBM bytes per launch of the round kernels from the committed rocprofv3 --pmc passes (scripts/gpu_pmc.sh), only if they
    were collected on the kernel set that is running: (the file's document: "kernels", optional "trace", file name) or (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_pmc_hbm.json")), reverse=True)
    for f in files:
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("engine_version") == engine_version:
            return pm, os.path.relpath(f, ROOT)
    return None, (f"no profiles/*_c3_pmc_hbm.json carries engine_version {engine_version}"
                  + (f" (newest: {os.path.basename(files[0])})" if files else ""))


def mfma_ceiling_tf(bf16_sweep, bf16_rtz):
    """fp32-exact matrix-pipe ceiling of a round whose two GEMMs (equal flops) ran on the given instructions: harmonic mean."""
    a = BF16X3_PEAK_TF if bf16_sweep else F32_MFMA_PEAK_TF
    b = BF16X3_PEAK_TF if bf16_rtz else F32_MFMA_PEAK_TF
    return 2.0 / (1.0 / a + 1.0 / b)


def roofline_block(config, N, d, B, K, ktimes, steps, rounds, wide, ktimes_all=None, bf16_sweep=False, bf16_rtz=False):
    """`roofline` of the dominant kernel of the configuration + per-family kernel milliseconds.

    C2 / C3 / C4 (K <= 112, d <= 64): k_round, one persistent launch per update_R sweep:
    HBM-bound, algorithmic bytes per cell 4d + 4K + 4 (Z_cos row, R row, list entry).
    C5 (wide shapes): k_assign_wide, one launch per update block: f32-MFMA-bound, 2 d K flop per cell."""
    import harmonypy_amd
    tot, cnt_timed = ktimes.get("assign_block", (0.0, 0))
    per_launch_ms = tot / max(cnt_timed, 1)
    n_rounds = steps * rounds
    # launches of the family in the timed region (the events bracket every TIMED_LAUNCH_STRIDE-th of them): from the
    # one-step pass that bracketed everything
    cnt = (ktimes_all["assign_block"][1] * steps) if ktimes_all and "assign_block" in ktimes_all else cnt_timed * TIMED_LAUNCH_STRIDE
    sweep = cnt <= n_rounds          # one launch per update_R sweep; else one launch per block
    cells_per_launch = N if sweep else N / 20.0
    fam_ms = {k: round(v[0], 3) for k, v in ktimes.items()}
    # kernel time of a whole round: the dominant kernel from the timed region, the other families of the round from the
    # one-step pass that bracketed every family (ktimes_all: one step = `rounds` rounds)
    t_round_kernels = per_launch_ms * (cnt / max(n_rounds, 1))
    if ktimes_all:
        t_round_kernels += sum(ktimes_all[k][0] for k in ("rtz_round", "rtz_reduce", "block_table") if k in ktimes_all) / max(rounds, 1)
    else:
        t_round_kernels += sum(ktimes[k][0] for k in ("rtz_round", "rtz_reduce", "block_table") if k in ktimes) / max(n_rounds, 1)
    round_traffic = None
    if wide:
        flops = cells_per_launch * 2.0 * d * K
        achieved = flops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        peak = BF16X3_PEAK_TF if bf16_sweep else F32_MFMA_PEAK_TF
        roof = {"bound": "mfma", "kernel": ("k_assign_wide3 (one launch per update block, centroids pre-split into bf16 fragments; K > 112 or d > 64)"
                                            if bf16_sweep else "k_assign_wide (one launch per update block, f32-input MFMA; K > 112 or d > 64)"),
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "frac_of_f32_mfma_peak": achieved / F32_MFMA_PEAK_TF,   # (comparable with rounds 1-4, whose `frac` was against 157.3)
                "peak_note": "fp32-exact ceiling of the instruction that ran: bf16x3 = 2.5 PFLOP/s / 6 products, f32-input MFMA = 157.3",
                "traffic": None, "traffic_source": "not collected for this configuration",
                "avg_launch_us": per_launch_ms * 1e3, "launches": cnt, "launches_timed": cnt_timed, "algorithmic_flops_per_launch": flops}
    else:
        alg_bytes = cells_per_launch * (4 * d + 4 * K + 4)
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        sweep_name = "k_round"
        kernel = (f"{sweep_name} (one persistent launch per update_R sweep: all 20 blocks)" if sweep
                  else "k_assign_lds (one launch per update block)")
        traffic, traffic_src, round_traffic, trace = None, None, None, None
        if sweep and config == "c3":
            pm, src = newest_pmc(harmonypy_amd.engine_version())
            traffic_src = src
            if pm is not None:
                round_traffic = 0.0
                for name, rec in pm["kernels"].items():
                    if name.startswith("void " + sweep_name):
                        traffic = rec["hbm_bytes_corrected"]
                    # the round's kernels: the sweep, the streaming R^T.Z pass over Z_cos (its one-hot block columns: <.., 1>), its finish kernel
                    if name.startswith("void " + sweep_name) or (name.startswith("void k_rtz3") and name.rstrip(">(Rtz3Args)").endswith("1")) or name.startswith("k_rtz3_finish"):
                        round_traffic += rec["hbm_bytes_corrected"]
                trace = pm.get("trace")
        roof = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": per_launch_ms * 1e3, "launches": cnt, "launches_timed": cnt_timed, "algorithmic_bytes_per_launch": alg_bytes}
        if trace and trace.get("k_round_avg_us"):
            # the same kernel in the rocprofv3 kernel trace of the build that is running (stored next to its counters):
            # a few per cent longer than between HIP events on the stream
            roof["frac_trace"] = alg_bytes / (trace["k_round_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            roof["trace"] = trace
    round_bytes = N * (4 * d + 8 * K + 8)
    round_flops = N * 4 * d * K
    # Which roof: both GEMMs of a round (distance product in the sweep, R^T.Z in the streaming pass) keep fp32 operands and
    # fp32 accumulators; HOW they multiply decides the matrix-pipe ceiling -- the f32-input MFMA (157.3 TFLOP/s) or six bf16
    # MFMAs per product on three exact bf16 terms per operand (2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-exact work).  The
    # round is priced against the instruction that RAN (the engine's counters say which).
    ceiling = mfma_ceiling_tf(bf16_sweep, bf16_rtz)
    t_hbm_us = round_bytes / (HBM_PEAK_GBS * 1e9) * 1e6
    t_mfma_us = round_flops / (ceiling * 1e12) * 1e6
    roof["round"] = {
        "algorithmic_bytes": round_bytes, "kernel_ms_per_round": t_round_kernels,
        # HBM bytes of the round's kernels (sweep + streaming R^T.Z pass + its finish kernel) from the committed counter passes
        # of the running build, and their ratio to the algorithmic bytes: re-reads show here
        "traffic": (round_traffic if not wide else None), "traffic_over_algorithmic": (round_traffic / round_bytes if (not wide and round_traffic) else None),
        "frac_of_hbm_peak": (round_bytes / (t_round_kernels * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_round_kernels > 0 else 0.0,
        "mfma_flops": round_flops,
        "mfma_ceiling_fp32_exact_tf": {"f32_input_mfma": F32_MFMA_PEAK_TF, "bf16x3": BF16X3_PEAK_TF, "this_run": ceiling,
                                       "instruction": {"sweep": "bf16x3" if bf16_sweep else "f32-input", "rtz": "bf16x3" if bf16_rtz else "f32-input"}},
        "frac_of_mfma_ceiling": (round_flops / (t_round_kernels * 1e-3) / (ceiling * 1e12)) if t_round_kernels > 0 else 0.0,
        "frac_of_f32_mfma_peak": (round_flops / (t_round_kernels * 1e-3) / (F32_MFMA_PEAK_TF * 1e12)) if t_round_kernels > 0 else 0.0,
        "floor_us_per_round": {"hbm_8TBs": t_hbm_us, "hbm_copy_ceiling_6.29TBs": t_hbm_us * 8.0 / 6.29, "mfma_this_run": t_mfma_us},
        "bound": "hbm" if t_hbm_us >= t_mfma_us else "mfma"}
    roof["engine_version"] = harmonypy_amd.engine_version()
    return roof, fam_ms


def side_config(name, rounds, steps, warmup, device, repeats=1, converge=False):
    """Throughput of another BASELINE configuration with the same step definition (single GPU).  `repeats` timed regions of
    `steps` steps each: the entry reports their median (and all of them).  `converge`: also a default run to convergence on
    the same cells (its own initialisation), wall-clock -- the second figure of BASELINE.json's metric."""
    from harmonypy_amd import harmony as H
    N, d, B, K = CONFIGS[name]
    Z, meta = synthetic_dataset(N, d, B, K, seed=0, cell_seed=0)
    ho = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, verbose=False, random_state=0, device=device,
                       _y0=quick_centroids(Z, K, seed=0))

    def step():
        ho.cluster(_rounds=rounds)
        ho.moe_correct_ridge()
        ho.check_convergence(1)
    for _ in range(warmup):
        step()
    dts = []
    for _ in range(max(1, repeats)):
        ho._engine.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ho._engine.sync()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    counters = ho._engine.counters()
    # whole-step roofline of the side entry from its wall time (no per-kernel events here): a round moves 4d + 8K + 8
    # algorithmic bytes and 4 d K flops per cell (distance product + R^T.Z), the ridge step 8K + 16d + 8 bytes and 4 d K flops
    # (statistics + correction) -- SURVEY section 8d's per-cell figures; the wide shapes are priced against the fp32-exact
    # ceiling of the instruction that ran (bf16x3: 417 TFLOP/s; f32-input MFMA: 157.3)
    t_step = dt / steps
    step_bytes = N * (rounds * (4.0 * d + 8 * K + 8) + 8.0 * K + 16 * d + 8)
    step_flops = N * (rounds + 1) * 4.0 * d * K
    wide = K > 112 or d > 64
    peak_tf = mfma_ceiling_tf(counters.get("sweeps_bf16_pipe", 0) > 0, counters.get("rtz_bf16_pipe", 0) > 0)
    roof = ({"bound": "mfma", "achieved": step_flops / t_step / 1e12, "peak": peak_tf, "unit": "TFLOP/s",
             "frac": step_flops / t_step / 1e12 / peak_tf, "frac_of_f32_mfma_peak": step_flops / t_step / 1e12 / F32_MFMA_PEAK_TF,
             "peak_note": "fp32-exact ceiling of the instructions that ran (harmonic mean over the round's two GEMMs): bf16x3 = 2.5 PFLOP/s / 6, f32-input MFMA = 157.3"}
            if wide else
            {"bound": "hbm", "achieved": step_bytes / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": step_bytes / t_step / 1e9 / HBM_PEAK_GBS})
    roof["scope"] = "whole step (wall time of the loop, launch gaps included)"
    out = {"workload": f"{config_label(name)}: {N} cells x {d} PCs, "
                       f"{B} batches, K={K}; step = {rounds} k-means rounds + 1 ridge correction",
           "value": N * steps / dt, "unit": "cells/sec/Harmony-iteration", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "roofline": roof}
    if wide:   # how the wide sweep ran (hmx_counters out[12]): all 20 blocks in one persistent launch, or one launch per block
        out["sweep"] = ("one persistent launch per round (k_sweep_wide3: block sums as self-validating fixed-point words)"
                        if counters.get("sweeps_wide_persistent", 0) > 0 else "one launch per block (k_assign_wide3)")
    if len(dts) > 1:
        out["repeats_ms_per_step"] = [round(1e3 * x / steps, 4) for x in dts]
        out["value_is"] = f"median of {len(dts)} timed regions of {steps} steps"
    if converge:
        del ho
        t0 = time.perf_counter()
        ho2 = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, verbose=False, random_state=0, device=device)
        ho2._engine.sync()
        t_init = time.perf_counter() - t0
        t1 = time.perf_counter()
        ho2.harmonize(10, verbose=False)                               # max_iter_harmony default (harmony.py:58)
        ho2._engine.sync()
        t_loop = time.perf_counter() - t1
        out["convergence"] = {"wall_s": t_init + t_loop, "setup_and_init_s": t_init, "harmonize_loop_s": t_loop,
                              "harmony_iterations": len(ho2.kmeans_rounds), "kmeans_rounds": [int(r) for r in ho2.kmeans_rounds],
                              "converged": bool(ho2.check_convergence(1)), "cells_total": N,
                              "setup_breakdown_s": {k: round(v, 4) for k, v in ho2.timing.items() if k != "harmonize"},
                              "init": "host arrays -> upload (rows regrouped on the GPU) + k-means++ seeds on a 32k-cell subsample (GPU) + 25 Lloyd "
                                      "iterations over all cells (GPU) + init_cluster, then harmonize() to convergence; wall-clock on ONE GPU"}
    return out


def side_config_fresh_process(name, rounds, steps, warmup, repeats):
    """The side entry measured in a process of its own (nothing of this process's allocations, LISI buffers or freed engines
    in front of it): `python bench.py --side NAME ...` prints the entry as one JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--side", name, "--rounds", str(rounds), "--steps", str(steps),
           "--warmup", str(warmup), "--side-repeats", str(repeats)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        out["measured_in"] = "a fresh process"
        return out
    except Exception as ex:   # fall back to this process, say so
        out = side_config(name, rounds, steps, warmup, "cuda:0", repeats=repeats)
        out["measured_in"] = f"this process (fresh process failed: {type(ex).__name__})"
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 on one GPU; c4 = BASELINE configs[3] (10M cells sharded over the ranks) on several")
    ap.add_argument("--rounds", type=int, default=10, help="k-means rounds per Harmony iteration")
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="cells of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-only", action="store_true", help="print only the cpu_baseline object (no GPU needed)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-lisi", action="store_true", help="skip the LISI of the embedding before / after the run to convergence")
    ap.add_argument("--lisi-cells", type=int, default=1_000_000, help="cells of the LISI measurement (evenly spaced subsample above that)")
    ap.add_argument("--no-convergence", action="store_true", help="skip the untimed end-to-end run to convergence")
    ap.add_argument("--side", default=None, choices=sorted(CONFIGS), help="print only the side entry of that configuration (one JSON line)")
    ap.add_argument("--side-repeats", type=int, default=3)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if args.side:
        os.environ["HMX_UPDATE_ORDER"] = "device"
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        entry = side_config(args.side, args.rounds, args.steps, args.warmup, "cuda:0", repeats=args.side_repeats)
        os.write(json_fd, (json.dumps(entry) + "\n").encode())
        return 0
    if args.cpu_only:
        N0, d0, B0, K0 = CONFIGS[args.config or "c3"]
        print(json.dumps(cpu_baseline(d0, B0, K0, args.rounds, min(args.cpu_sample, N0))))
        return 0
    # --gpus N from a plain command line: start the N ranks ourselves (one process per GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        return subprocess.call(self_launch(sys.argv[1:], args.gpus))
    # stdout carries exactly one JSON line: libraries that print banners through C stdio (RCCL's
    # version banner, for one) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a figure for a different GPU count", file=sys.stderr)
        return 2
    import torch
    backend = os.environ.get("HMX_BENCH_BACKEND", "nccl")   # gloo: several ranks on one GPU (tests of the sharded path)
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (backend == "nccl" and n_dev < world):
        print(f"bench.py: {world} rank(s) need {world} GPU(s), {n_dev} visible", file=sys.stderr)
        return 3
    dist = None
    shard = None
    if world > 1 or os.environ.get("HMX_BENCH_FORCE_SHARD"):
        import torch.distributed as dist
        local_rank = local_rank % n_dev   # several ranks on one GPU only happen in tests
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend)

    config, N, scaling = resolve_workload(args.config, world)
    _, d, B, K = CONFIGS[config]
    if os.environ.get("HMX_BENCH_CELLS"):                   # tests: a smaller job, same shape
        N = int(os.environ["HMX_BENCH_CELLS"])
    os.environ["HMX_UPDATE_ORDER"] = "device"
    import harmonypy_amd
    from harmonypy_amd import harmony as H

    # every rank holds one shard of N cells of ONE job of world*N cells: the same cell types and batch effects on
    # every rank (seed 0), different cells (cell_seed = rank)
    Z, meta = synthetic_dataset(N, d, B, K, seed=0, cell_seed=rank)
    Y0 = quick_centroids(Z, K, seed=0) if rank == 0 else None
    if dist is not None:
        shard = harmonypy_amd.Shard()            # nccl group -> the engine's own RCCL communicator
        Y0 = shard.broadcast_object(Y0)
    t_setup = time.perf_counter()
    ho = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, verbose=False, random_state=0,
                       device=f"cuda:{local_rank}", shard=shard, _y0=Y0)
    t_setup = time.perf_counter() - t_setup

    def step():
        ho.cluster(_rounds=args.rounds)
        ho.moe_correct_ridge()
        try:
            ho.check_convergence(1)
        except ZeroDivisionError:   # only with the timing-experiment builds (HMX_LIB), whose sums are void
            pass

    def fence():
        ho._engine.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timing = not args.no_roofline
    ktimes_all = {}
    if timing:
        # every family, one untimed step: the per-family kernel milliseconds of the line (diagnosis).  Bracketing all
        # launches with events costs 11 % of a step at C3, so the timed region below brackets the dominant kernel only.
        ho._engine.enable_timing(True)
        step()
        ktimes_all = ho._engine.kernel_times()
        # ... and of its launches every fourth: a pair of event records costs ~15 us of queue time, 5 % of a C3 round
        ho._engine.enable_timing(True, families=["assign_block"], stride=TIMED_LAUNCH_STRIDE)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if "nccl" in dist.get_backend() else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = ho._engine.kernel_times() if timing else {}
    ho_wide = ho._wide_shape()
    counters = ho._engine.counters()
    if timing:
        ho._engine.enable_timing(False)
    transport = str(getattr(ho, "transport", None))
    # every rank's view of the sharded path, for diagnosis of a scaling run: transport, whether the in-kernel peer
    # exchange passed its self-test there, the hops of the sweep kernel's grid-wide waits, collectives issued
    per_rank = None
    if shard is not None:
        mine = {"rank": rank, "device": local_rank, "transport": transport, "peer_exchange": bool(getattr(shard, "peer_exchange", False)),
                "sweep_waits": counters["sweep_waits"], "incomplete_polls": counters["sweep_wait_polls"],
                "incomplete_polls_max": counters["sweep_wait_polls_max"], "fallback_rounds": counters["sweep_fallbacks"],
                "collectives": counters["collectives"], "timed_s": dt}
        per_rank = shard.allgather_object(mine)

    # ---- second figure of BASELINE.json's metric: wall-clock to convergence of a default run_harmony on the
    #      same cells (its own initialisation: k-means++ seeds on a subsample + Lloyd iterations on the GPU,
    #      natural round / iteration counts), outside the timed region of `value`
    conv = None
    if not args.no_convergence:
        del ho
        fence_t = time.perf_counter()
        ho2 = H.run_harmony(Z, meta, ["batch"], nclust=K, max_iter_harmony=0, verbose=False, random_state=0,
                            device=f"cuda:{local_rank}", shard=shard)
        ho2._engine.sync()
        t_init = time.perf_counter() - fence_t
        t1 = time.perf_counter()
        ho2.harmonize(10, verbose=False)                               # max_iter_harmony default (harmony.py:58)
        ho2._engine.sync()
        t_loop = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([t_init, t_loop], device="cuda" if "nccl" in dist.get_backend() else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_init, t_loop = (float(x) for x in tt.tolist())
        conv = {"wall_s": t_init + t_loop, "setup_and_init_s": t_init, "harmonize_loop_s": t_loop,
                "harmony_iterations": len(ho2.kmeans_rounds), "kmeans_rounds": [int(r) for r in ho2.kmeans_rounds],
                "converged": bool(ho2.check_convergence(1)), "cells_total": N * world,
                "setup_breakdown_s": {k: round(v, 4) for k, v in ho2.timing.items() if k != "harmonize"},
                "init": "upload (rows regrouped on the GPU) + k-means++ seeds on a 32k-cell subsample (GPU) + 25 Lloyd "
                        "iterations over all cells (GPU) + init_cluster; not part of `value`"}
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    ms_per_step = 1e3 * dt / args.steps
    value = N * world * args.steps / dt
    job = (f"{config_label(config)}: " +
           (f"{N * world} cells x {d} PCs, {B} batches, K={K} sharded over {world} GPUs ({N} cells per GPU)" if scaling == "strong"
            else f"{N} cells x {d} PCs, {B} batches, K={K} per GPU"))
    import hashlib
    import socket
    out = {
        "metric": "cells/sec/Harmony-iteration", "value": value, "unit": "cells/sec/Harmony-iteration",
        # which machine: the same build measures 5-6 % apart on different boxes of the pool, so a
        # difference between two lines is progress only if their `box` agrees or an A/B on ONE box backs it (profiles/*ab*)
        "box": hashlib.sha256(socket.gethostname().encode()).hexdigest()[:8],
        "box_to_box_spread_note": "same build: 5-6 % between boxes; gains are claimed from same-box A/B runs (profiles/r0*_ab_*.txt)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{job}; step = 1 Harmony iteration = "
                        f"{args.rounds} k-means rounds (block_size 0.05 -> 20 blocks) + 1 ridge correction",
            "cells_per_gpu": N, "pcs": d, "batches": B, "clusters": K, "rounds_per_iteration": args.rounds,
            "update_order": "device (keyed bijection, generated inside the timed region)",
            # fp32 operands, fp32 accumulators, fp32 results; HOW the sweep multiplies (tests/test_split_gemm.py)
            "distance_gemm": ("bf16x3: every fp32 operand as the exact sum of three bf16 terms, six bf16 MFMAs per 32 k with fp32 "
                              "accumulation (as close to float64 as the f32-input MFMA)" if counters.get("sweeps_bf16_pipe", 0) > 0
                              else "f32-input MFMA"),
            # how the sweep kernel deals a block's tiles out and hands the block sums on
            "sweep_tile_map": (f"group-affine: one batch group per workgroup ({counters.get('sweep_group_affine_wgs', 0)} workgroups), block sums as "
                               "self-validating fixed-point words (count << 55 | sum * 2^32)" if counters.get("sweeps_group_affine", 0) > 0
                               else "classic: tile pairs round-robin over all workgroups, returning fp64 adds + arrival counter"),
            "init": "k-means++ on a 50k-cell subsample, untimed",
            "parallelism": (f"cells sharded over {world} ranks (1 per GPU), transport {transport}: "
                            + ("block sums exchanged inside the sweep kernel through peer boxes (xGMI), 1 all-reduce per "
                               "round + 1 per ridge" if "+peer" in transport else
                               "1 + 20 all-reduces per round, 1 per ridge")) if shard is not None else "single GPU",
            "cells_total": N * world,
            "cell_rounds_per_sec": N * world * args.steps * args.rounds / dt,
            "setup_s": t_setup,
        },
    }
    # the grid-wide waits of the sweep kernel on this rank (the hop every block pays; across ranks when sharded): how often
    # a poll found the hand-off incomplete (each followed by ~64 shader cycles of sleep) -- rank 0's view, for diagnosis
    out["sweep_waits"] = {"waits": counters["sweep_waits"], "incomplete_polls_mean": counters["sweep_wait_polls"] / max(counters["sweep_waits"], 1),
                          "incomplete_polls_max": counters["sweep_wait_polls_max"], "fallback_rounds": counters["sweep_fallbacks"],
                          "collectives": counters["collectives"]}
    if per_rank is not None:
        polls = [r["incomplete_polls"] / max(r["sweep_waits"], 1) for r in per_rank]
        out["ranks"] = {"per_rank": per_rank,
                        "incomplete_polls_mean": {"min": min(polls), "mean": sum(polls) / len(polls), "max": max(polls)},
                        "peer_exchange_on_all": all(r["peer_exchange"] for r in per_rank),
                        "transports": sorted(set(r["transport"] for r in per_rank)),
                        "collectives_per_rank": sorted(set(r["collectives"] for r in per_rank))}
    if timing:
        out["roofline"], _ = roofline_block(config, N, d, B, K, ktimes, args.steps, args.rounds, ho_wide, ktimes_all,
                                            bf16_sweep=counters.get("sweeps_bf16_pipe", 0) > 0, bf16_rtz=counters.get("rtz_bf16_pipe", 0) > 0)
        out["kernel_ms_per_step"] = {k: round(v[0], 3) for k, v in ktimes_all.items()}
    if conv is not None:
        out["convergence"] = conv
    if conv is not None and world == 1 and not args.no_lisi:
        # the reference's integration metric (lisi.py) on the same device: batch mixing of the embedding
        # before and after the run to convergence above (perplexity 30 -> 89 exact neighbours per cell)
        take = np.arange(N) if N <= args.lisi_cells else np.linspace(0, N - 1, args.lisi_cells).astype(np.int64)
        meta_l = meta.iloc[take]
        harmonypy_amd.compute_lisi(Z[take[:4096]], meta_l.iloc[:4096], ["batch"], 30, device=f"cuda:{local_rank}")   # warm-up
        t_l = time.perf_counter()
        before = harmonypy_amd.compute_lisi(Z[take], meta_l, ["batch"], 30, device=f"cuda:{local_rank}")
        t_l = time.perf_counter() - t_l
        Zc_take = ho2.Z_corr[take]
        del ho2                                                     # (the side configurations below get the GPU to themselves)
        after = harmonypy_amd.compute_lisi(Zc_take, meta_l, ["batch"], 30, device=f"cuda:{local_rank}")
        out["lisi"] = {"cells": int(len(take)), "pcs": d, "perplexity": 30, "seconds": t_l, "cells_per_sec": len(take) / t_l,
                       "batches": B, "batch_lisi_before": float(before.mean()), "batch_lisi_after": float(after.mean()),
                       "note": "host float64 input to host output, exact neighbours; not part of `value`"}
    if world == 1 and config == "c3" and not args.no_convergence:
        # BASELINE configs[1] (69k cells x 50 PCs, 4 batches, K=30) measured the same way, for reference: it is
        # latency-bound (its working set lives in the L3; 20 sequential hand-offs per round), so the headline
        # figure is quoted on configs[2], the roofline point
        # (in a process of its own, median of three timed regions: inside this process the entry read 15 % low in round 4)
        del Z, meta
        out["configs_1"] = side_config_fresh_process("c2", args.rounds, steps=20, warmup=5, repeats=3)
        # all 10M cells of BASELINE configs[3] on this ONE GPU (the N=1 point of the strong-scaling curve `--gpus N`
        # measures) incl. the default run to convergence on them (north_star: "10M cells converged in < 10 s"), and the
        # per-GPU shard of configs[4] on 8 GPUs (wide-PC regime)
        out["configs_3_on_one_gpu"] = side_config("c4x1", args.rounds, steps=2, warmup=1, device=f"cuda:{local_rank}", converge=True)
        out["configs_3_half_on_one_gpu"] = side_config("c4x2", args.rounds, steps=3, warmup=1, device=f"cuda:{local_rank}")
        # (a process of its own, median of three regions: behind the 10 M-cell entries of this process the same build read 55.6-59.4 M)
        out["configs_4_shard"] = side_config_fresh_process("c5", args.rounds, steps=2, warmup=1, repeats=3)
    if args.cpu_sample > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(d, B, K, args.rounds, min(args.cpu_sample, N))
        # the survey's own measurement of the reference at the HEADLINE size (the CPU path is super-linear in N: the sample
        # above flatters it): context beside the in-run figure, never a target
        out["cpu_baseline"]["reference_at_1M_survey"] = {"value": REFERENCE_AT_1M_SURVEY, "unit": "cells/sec/Harmony-iteration", "cores": 8,
                                                        "source": "BASELINE.md section 2: harmonypy run_harmony(device='cpu') at 1 M cells x 50 PCs, K=100, measured by the survey (not in this run)"}
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
