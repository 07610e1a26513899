#!/usr/bin/env python3
"""BASELINE config 4: ONE exact GP sharded over the GPUs of a node (column-panel block-cyclic Cholesky).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 \
        tools/run_sharded.py --n 262144 --nb 512

Prints one JSON line on rank 0 (fit / predict wall-clock, GFLOP/s, LML, first posterior means)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--m", type=int, default=300)
    ap.add_argument("--nb", type=int, default=512)
    args = ap.parse_args()
    import torch

    from battgp_amd import KERNEL_BATTGP, parallel, synthetic
    from battgp_amd.sharded import make_sharded_gp

    gp = make_sharded_gp(KERNEL_BATTGP, synthetic.HYP_BATTGP, nb=args.nb)
    x, y = synthetic.make_cell_data(args.n)
    xq = synthetic.make_query(x, args.m)
    parallel.barrier(gp.dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lml = gp.fit(x, y)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mean, var = gp.predict(xq)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fit_s = parallel.max_over_ranks(gp.dist, t1 - t0, device=gp.be.device)
    pred_s = parallel.max_over_ranks(gp.dist, t2 - t1, device=gp.be.device)
    if gp.rank == 0:
        n, m = args.n, args.m
        flop = n**3 / 3.0 + float(n) * n * m + 2.0 * n * n
        print(json.dumps({
            "workload": f"full_gp sharded, N={n}, nb={args.nb}, world={gp.world}",
            "fit_s": fit_s, "predict_s": pred_s, "gflops": flop / (fit_s + pred_s) / 1e9,
            "lml": lml, "jitter": gp.jitter, "mean_first": [float(v) for v in mean[:3]], "var_first": [float(v) for v in var[:3]],
        }), flush=True)
    parallel.barrier(gp.dist)
    dist = gp.dist
    gp.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
