#!/usr/bin/env python3
"""bench.py - exact-GP fit+predict on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic cell: covariance fill over
(time, I, SOC, T) inputs -> jittered blocked Cholesky -> z, alpha, LML -> cross fill,
posterior mean and variance at M = 300 query points.  Inputs (X, y, Xq) are resident in HBM
before the timed region starts; results stay on the device.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 40000] [--kernel battgp]

N > 1 is launched by torch.distributed.run, one rank per GPU: every rank fits its OWN cell
(the reference's "8-cell pack" = independent GPs, src/batt_models/battgp_full.py:41-60), no
data-path collective; barrier + max-over-ranks timing; value = whole-job GFLOP/s.
Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak (= 256 CU x 2.4 GHz x 128 flop/clk)
PEAK_HBM_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_flop(n: int, m: int) -> float:
    """N^3/3 (Cholesky) + N^2 M (predictive TRSM) + 2 N^2 (two TRSV)."""
    return n**3 / 3.0 + float(n) * n * m + 2.0 * n * n


def cpu_baseline(kernel_id, hyp, n_cpu: int, m: int, seed: int):
    """Oracle (numpy fill + LAPACK dpotrf/dtrtrs) timed on the host cores: baseline only."""
    from threadpoolctl import threadpool_info

    from battgp_amd import synthetic
    from oracle.exact_gp import OracleGP

    x, y = synthetic.make_cell_data(n_cpu, seed=seed)
    xq = synthetic.make_query(x, m)
    t0 = time.perf_counter()
    gp = OracleGP(kernel_id, hyp, x, y).fit()
    gp.predict(xq)
    dt = time.perf_counter() - t0
    threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    # BASELINE configs[0] (N = 2048, the reference's own CPU-runnable case) beside it
    x0, y0 = synthetic.make_cell_data(2048, seed=2048)
    xq0 = synthetic.make_query(x0, m)
    t0 = time.perf_counter()
    OracleGP(kernel_id, hyp, x0, y0).fit().predict(xq0)
    dt0 = time.perf_counter() - t0
    return {
        "config0_n2048": {"seconds": dt0, "value": algorithmic_flop(2048, m) / dt0 / 1e9},
        "value": algorithmic_flop(n_cpu, m) / dt / 1e9,
        "unit": "GFLOP/s",
        "cores": int(threads),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "seconds": dt,
        "sample": f"same workload at N={n_cpu} (one fit+predict, M={m}); numpy fill + LAPACK dpotrf/dtrtrs via scipy/OpenBLAS",
    }


def traffic_from_profile(n: int, kernel: str, key: str = "hbm_bytes_per_dispatch"):
    """Per-dispatch HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) or PMC MFMA utilisation of the trailing-update
    kernel from the committed rocprofv3 passes (tools/profile_round.sh -> profiles/*_summary.json, which
    holds the calibration); None for workloads that were not profiled."""
    path = os.path.join(ROOT, "profiles", f"r01_n{n}_summary.json")
    if kernel != "battgp" or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["gemm_nt_128x128"].get(key)


def n_max_from_profile():
    """Largest N measured on one MI355X (column-slab layout of the factor, tools/large_n.py): the committed
    record, not re-measured here (one fit at that size takes ~100 s)."""
    path = os.path.join(ROOT, "profiles", "r01_large_n.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        runs = json.load(f)["runs"]
    best = max(runs, key=lambda r: (r["n"], r.get("panel_scheme", 0)))
    return {
        "n": best["n"], "fit_predict_s": best["fit_predict_s"], "gflops": best["gflops"], "slab_width": best["slab_width"],
        "factor_bytes": best["factor_bytes"], "residuals": best["residuals"], "full_square_limit_n": 196000,
        "source": "profiles/r01_large_n.json (tools/large_n.py; not re-measured by this run)",
    }


def target_size_report(n: int, m: int) -> dict:
    """One fit+predict per kernel at the size the north-star targets are quoted on (N = 131 072):
    fill GB/s vs 8 TB/s, trailing-update TFLOP/s vs 78.6, and on-device residuals as correctness
    evidence where no CPU oracle can follow.  Not part of `value`."""
    import torch

    from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic
    from battgp_amd.engine import EngineError, ExactGPEngine

    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, m)
    rep = {"n": n}
    for name, kid, hyp in (("battgp", KERNEL_BATTGP, synthetic.HYP_BATTGP), ("matern32", KERNEL_MATERN32, synthetic.HYP_MATERN32)):
        eng = ExactGPEngine(kid, hyp, device=torch.cuda.current_device())
        try:
            # fill kernel alone, launched back to back on a scratch matrix of the same shape: its steady
            # rate.  (Inside a fit the fill is the first big kernel after a host-side gap and pays ~2 ms of
            # clock ramp-up - tools/fill_hot_probe.py - which is reported separately as fill_gbs.)
            ld = n + 384
            scratch = torch.empty((n, ld), dtype=torch.float64, device="cuda")
            tx = torch.from_numpy(x).cuda()
            torch.cuda.synchronize()
            rates = []
            for _ in range(8):
                eng.fill_device(tx.data_ptr(), n, tx.data_ptr(), n, 4, scratch.data_ptr(), ld, lower=1, diag_add=float(hyp[0]))
                ph = eng.phase_times()
                rates.append(ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9)
            del scratch, tx
            torch.cuda.empty_cache()
            fill_steady = float(np.median(rates[4:]))  # the first launches still see the clocks ramp up
            eng.fit_predict(x, y, xq)  # first pass from an idle, down-clocked GPU: warm-up only
            t0 = time.perf_counter()
            eng.fit_predict(x, y, xq)  # fill + factorisation with the query rows riding + posterior
            wall = time.perf_counter() - t0
            ph = eng.phase_times()
            res = eng.residuals(256)
            rep[name] = {
                "fit_predict_s": wall,
                "fill_gbs": ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9,
                "fill_frac_hbm": ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "fill_steady_gbs": fill_steady,
                "fill_steady_frac_hbm": fill_steady / PEAK_HBM_GBS,
                "potrf_tflops": (n**3 / 3.0) / (ph["potrf_ms"] * 1e-3) / 1e12,
                "trail_tflops": ph["trail_flop"] / (ph["trail_ms"] * 1e-3) / 1e12,
                "trail_frac_mfma": ph["trail_flop"] / (ph["trail_ms"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                "phases_ms": {k: ph[k] for k in ("fill_ms", "potrf_ms", "solve_ms", "cross_ms", "var_ms", "trail_ms")},
                "lml": eng.lml,
                "jitter": eng.jitter,
                "residuals": {"rel_solve": res[0], "max_llt": res[1]},
                "device_bytes": eng.device_bytes(),
            }
        except EngineError as exc:  # e.g. not enough HBM on a smaller part
            rep[name] = {"error": str(exc)}
        finally:
            eng.close()
    return rep


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--size", dest="n", type=int, default=40000,
                    help="training points per cell (configs[1]: 40 000); use --size under torch.distributed.run, whose argparse rejects --n as an ambiguous prefix of its own options")
    ap.add_argument("--m", type=int, default=300, help="query points (battgp_full.py:98)")
    ap.add_argument("--kernel", default="battgp", choices=["battgp", "matern32"])
    ap.add_argument("--nb", type=int, default=-1, help="outer panel width override")
    ap.add_argument("--lookahead", type=int, default=-1, help="bits 0-2: look-ahead depth (0 off, 1 default); +8: panel-stream updates ordered before rest(k); +16: no atomic epilogue")
    ap.add_argument("--panel-scheme", type=int, default=-1, help="0 = 64-wide chain over all rows, 1 = diagonal-block chain + one deep TRSM GEMM (default)")
    ap.add_argument("--slab", type=int, default=0, help="bgp_set_layout: 0 automatic, -1 full square, > 0 column-slab width")
    ap.add_argument("--cpu-n", type=int, default=8192, help="size of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-residuals", action="store_true")
    ap.add_argument("--separate", action="store_true", help="bgp_fit then bgp_predict (separate triangular-solve pass) instead of the fused call")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1: nccl (= RCCL, the driver's runs); gloo with --share-gpu rehearses "
                         "the multi-rank path on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--target-n", type=int, default=131072,
                    help="also report the kernels' roofline fractions at the north-star size (1 GPU only; 0 = skip)")
    args = ap.parse_args()

    import torch

    from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic
    from battgp_amd.engine import ExactGPEngine

    from battgp_amd import parallel

    rank, world, local_rank = parallel.env_rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(
            f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})"
        )
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = parallel.init(args.backend, device=torch.device("cuda", local_rank))  # nccl = RCCL; None for 1 process
    red_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    kernel_id, hyp = (
        (KERNEL_BATTGP, synthetic.HYP_BATTGP) if args.kernel == "battgp" else (KERNEL_MATERN32, synthetic.HYP_MATERN32)
    )
    n, m = args.n, args.m
    # every rank = a different cell (different seed), same size: weak scaling, no collective
    x, y = synthetic.make_cell_data(n, seed=n + rank)
    xq = synthetic.make_query(x, m)
    dev = torch.device("cuda", local_rank)
    tx = torch.from_numpy(x).to(dev)
    ty = torch.from_numpy(y).to(dev)
    txq = torch.from_numpy(xq).to(dev)
    tmean = torch.empty(m, dtype=torch.float64, device=dev)
    tvar = torch.empty(m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    eng = ExactGPEngine(kernel_id, hyp, device=local_rank)
    if args.nb > 0:
        eng.set_options(nb_outer=args.nb)
    if args.lookahead >= 0:
        eng.set_options(lookahead=args.lookahead)
    if args.panel_scheme >= 0:
        eng.set_panel_scheme(args.panel_scheme)
    if args.slab != 0:
        eng.set_layout(args.slab)

    def step():
        if args.separate:
            eng.fit_device(tx.data_ptr(), ty.data_ptr(), n, 4)
            eng.predict_device(txq.data_ptr(), m, tmean.data_ptr(), tvar.data_ptr(), 1e-10)
        else:  # the reference's flow: the first predict triggers the factorisation; one fused pass
            eng.fit_predict_device(tx.data_ptr(), ty.data_ptr(), n, 4, txq.data_ptr(), m, tmean.data_ptr(), tvar.data_ptr(), 1e-10)

    def barrier():
        parallel.barrier(dist)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phases = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # every C-ABI call returns only after its own stream has drained
        phases.append(eng.phase_times())
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(dist, elapsed, device=red_dev)

    # The la(k) and rest(k) launches of a panel overlap each other in the default schedule (fastest wall
    # clock), which stretches both event-timed durations.  One extra, untimed step with la(k) ordered before
    # rest(k) (lookahead = 1 | 8, ~1 % slower overall) gives the kernel's own rate per launch.
    serial = None
    if args.lookahead < 0:
        eng.set_options(lookahead=1 | 8)
        step()
        ph2 = eng.phase_times()
        serial = ph2["trail_flop"] / (ph2["trail_ms"] * 1e-3) / 1e12 if ph2["trail_ms"] > 0 else None
        eng.set_options(lookahead=1)

    resid = None
    if not args.no_residuals:
        resid = eng.residuals(256)  # on-device correctness evidence at the benchmarked size
    mean_host = tmean.cpu().numpy()
    lml, jitter = eng.lml, eng.jitter
    mem_bytes = eng.device_bytes()
    eng.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flop = algorithmic_flop(n, m)
        value = world * flop * args.steps / elapsed / 1e9
        avg = {k: float(np.mean([p[k] for p in phases])) for k in phases[0]}
        trail_tflops = avg["trail_flop"] / (avg["trail_ms"] * 1e-3) / 1e12 if avg["trail_ms"] > 0 else 0.0
        fill_gbs = avg["fill_bytes"] / (avg["fill_ms"] * 1e-3) / 1e9 if avg["fill_ms"] > 0 else 0.0
        potrf_tflops = (n**3 / 3.0) / (avg["potrf_ms"] * 1e-3) / 1e12 if avg["potrf_ms"] > 0 else 0.0
        out = {
            "metric": "exact-GP fit+predict throughput (fp64)",
            "value": value,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"full_gp, {'Wiener+ARD-RBF (reference full_gp kernel)' if args.kernel == 'battgp' else 'Matern-3/2 ARD'}, "
                f"N={n} synthetic 4-D inputs, M={m} queries, one cell per GPU",
                "n": n,
                "m": m,
                "kernel": args.kernel,
                "flop_per_step": flop,
                "parallelism": f"{world} independent cells" if world > 1 else "1 cell",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "gemm_nt_kernel<128,128,2> (rank-NB SYRK trailing update of the blocked Cholesky)",
                "achieved": trail_tflops,
                "peak": PEAK_FP64_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": trail_tflops / PEAK_FP64_MFMA_TFLOPS,
                "achieved_while_running": (avg["trail_flop"] / (avg["trail_union_ms"] * 1e-3) / 1e12) if avg.get("trail_union_ms", 0) > 0 else None,
                "frac_while_running": (avg["trail_flop"] / (avg["trail_union_ms"] * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if avg.get("trail_union_ms", 0) > 0 else None,
                "achieved_non_overlapped": serial,
                "frac_non_overlapped": (serial / PEAK_FP64_MFMA_TFLOPS) if serial else None,
                "traffic": traffic_from_profile(n, args.kernel),
                "launches_per_step": None,
                "mfma_util_pmc": traffic_from_profile(n, args.kernel, "mfma_util"),
                "note": "sum of algorithmic flop m(m+1)k of the outer trailing updates / sum of their HIP-event durations "
                        "(= flop per launch / average launch duration); with look-ahead the la and rest launches of a panel "
                        "overlap each other and the next panel's factorisation, so this under-states the kernel "
                        "(achieved_while_running: the same flop over the UNION of the launch intervals of the timed steps, i.e. the rate "
                        "while at least one trailing update is running; achieved_non_overlapped: the same launches in one extra "
                        "untimed step with la(k) ordered before rest(k)); "
                        "mfma_util_pmc is SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) of the same kernel "
                        "from the serialised counter pass in profiles/",
            },
            "roofline_fill": {
                "bound": "hbm",
                "kernel": "fill_kernel (lower triangle, 4N(N+1) algorithmic bytes)",
                "achieved": fill_gbs,
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": fill_gbs / PEAK_HBM_GBS,
            },
            "phases_ms": {k: avg[k] for k in ("h2d_ms", "fill_ms", "potrf_ms", "solve_ms", "cross_ms", "var_ms", "d2h_ms", "trail_ms")},
            "potrf_tflops": potrf_tflops,
            "potrf_frac_of_peak": potrf_tflops / PEAK_FP64_MFMA_TFLOPS,
            "lml": lml,
            "jitter": jitter,
            "residuals": {"rel_solve": resid[0], "max_llt": resid[1]} if resid else None,
            "mean_first": [float(v) for v in mean_host[:3]],
            "device_bytes": mem_bytes,
            "n_max_per_gpu": n_max_from_profile(),
        }
        out["roofline"]["launches_per_step"] = int(round(avg["trail_launches"]))
        if world == 1 and args.target_n > 0 and args.target_n != n:
            out["target_size"] = target_size_report(args.target_n, m)
        if args.cpu_n > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(kernel_id, hyp, args.cpu_n, m, seed=args.cpu_n)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
