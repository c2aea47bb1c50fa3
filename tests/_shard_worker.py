"""One rank of a sharded run; launched by tests/test_sharded*.py as
``python tests/_shard_worker.py <mode> <case> <outdir> [options]`` with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment (gloo rendezvous on 127.0.0.1).

modes
  oracle   the cell-sharded CPU oracle (oracle/sharded_oracle.py) behind the product's own
           sharded front end (harmonypy_amd.harmony._prepare_inputs + dist.Shard)     [CPU]
  engine   harmonypy_amd.run_harmony(..., shard=Shard(transport=<opt>)) on the GPU     [GPU]
Every rank writes <outdir>/rank<r>.npz with its slice of Z_corr and the objective history.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, case, outdir = sys.argv[1:4]
    opts = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
    import torch
    import torch.distributed as dist
    from conftest import load_case
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    data, meta, vars_use, kw, g = load_case(case)
    kw.update(opts.get("kw", {}))
    N = data.shape[0]
    # uneven contiguous slices (rank 0 gets the smaller one): exercises offsets and ragged blocks
    cuts = np.linspace(0, N, world + 1).astype(int)
    if world > 1:
        cuts[1] = max(1, cuts[1] - N // 7)
    lo, hi = cuts[rank], cuts[rank + 1]
    Z_loc, meta_loc = data[lo:hi], meta.iloc[lo:hi].reset_index(drop=True)
    rounds = [int(r) for r in g["kmeans_rounds"]] if opts.get("forced", True) else None
    rs = kw.get("random_state", 0)

    from harmonypy_amd import Shard
    out = {}
    if mode == "oracle":
        from harmonypy_amd.harmony import _prepare_inputs
        from oracle.sharded_oracle import ShardedOracle
        shard = Shard(transport="host")
        run_kw = {k: kw[k] for k in ("theta", "lamb", "sigma", "nclust", "tau") if k in kw}
        p = _prepare_inputs(Z_loc, meta_loc, vars_use, shard=shard, **run_kw)
        torch.manual_seed(rs)
        oo = ShardedOracle(p["Z"], p["codes"].dense(), p["Pr_b"], p["sigma"], p["theta"], p["lamb"],
                           reduce=shard.allreduce_, offset=shard.offset, n_global=shard.n_global,
                           alpha=kw.get("alpha", 0.2), lambda_estimation=p["lambda_estimation"], K=p["K"],
                           block_size=kw.get("block_size", 0.05), forced_rounds=rounds,
                           max_iter_harmony=kw.get("max_iter_harmony", 10),
                           max_iter_kmeans=kw.get("max_iter_kmeans", 20),
                           epsilon_kmeans=kw.get("epsilon_cluster", 1e-5),
                           epsilon_harmony=kw.get("epsilon_harmony", 1e-4))
        oo.init_cluster(rs, g["Y0"])
        oo.harmonize(oo.max_iter_harmony)
        out = dict(Z_corr=oo.result(), objective_kmeans=oo.objective_kmeans, objective_harmony=oo.objective_harmony,
                   kmeans_rounds=oo.kmeans_rounds, K=p["K"], Pr_b=p["Pr_b"], theta=p["theta"],
                   n_collectives=shard.n_collectives)
    elif mode == "engine":
        from harmonypy_amd import harmony as H
        if opts.get("order"):
            os.environ["HMX_UPDATE_ORDER"] = opts["order"]
        shard = Shard(transport=opts.get("transport", "host"))
        ho = H.run_harmony(Z_loc, meta_loc, vars_use, verbose=False, shard=shard,
                           _y0=g["Y0"] if opts.get("Y0", True) else None, _schedule=rounds, **kw)
        out = dict(Z_corr=ho.Z_corr, objective_kmeans=ho.objective_kmeans, objective_harmony=ho.objective_harmony,
                   kmeans_rounds=ho.kmeans_rounds, K=ho.K, Pr_b=ho.Pr_b, theta=ho.theta, O=ho.O, E=ho.E, Y=ho.Y,
                   R_colsum_local=ho.R.sum(axis=0), transport=str(ho.transport),
                   sweep_fallbacks=ho._engine.counters()["sweep_fallbacks"], peer_box=ho._engine.counters()["peer_box"])
    else:
        raise SystemExit(f"unknown mode {mode}")
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
