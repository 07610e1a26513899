"""Hyper-parameter training through the reference-shaped surface (BatteryCellGP_Full.train_hyperparameters ->
training.train_exact_gp_adam, src/gp/training.py:11-67): wall clock per optimiser iteration.
    python tools/train_probe.py [N ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from battgp_amd import synthetic  # noqa: E402
from battgp_amd.battcellgp_full import BatteryCellGP_Full  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [1000, 4000]:
    x, y = synthetic.make_cell_data(n, seed=3)
    cell = BatteryCellGP_Full(x, y, cellnr=1, device=0, max_iter=40, rel_tol=0.0)
    t0 = time.perf_counter()
    losses = cell.train_hyperparameters(messages=False)
    dt = time.perf_counter() - t0
    print(json.dumps({"n": n, "iterations": int(len(losses)), "total_ms": dt * 1e3, "per_iteration_ms": dt * 1e3 / max(1, len(losses)),
                      "loss_first": float(losses[0]), "loss_last": float(losses[-1])}), flush=True)
    del cell.model
