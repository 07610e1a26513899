#!/usr/bin/env python3
"""bench.py - exact-GP fit+predict on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic cell: covariance fill over (time, I, SOC, T) inputs ->
jittered blocked Cholesky -> z, LML -> posterior mean and variance at M = 300 query points (the query rows ride
through the factorisation).  Inputs (X, y, Xq) are resident in HBM before the timed region starts; results stay on
the device.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 131072] [--kernel matern32]

Headline workload (N = 1): BASELINE configs[2], the largest single-GPU configuration - full_gp, Matern-3/2 + noise,
N = 131 072 - the size the north-star's kernel targets are quoted on.  configs[1] (N = 40 000, the reference kernel)
rides along as `extra_configs`.

N > 1 runs one rank per GPU under torch.distributed.run (the driver's launch); a bare `python bench.py --gpus N` launches
itself that way (self_launch):
  --mode cells   (default)  every rank fits its OWN cell (the reference's "8-cell pack" = independent GPs,
                 src/batt_models/battgp_full.py:41-60); no data-path collective; weak scaling; value = whole-job GFLOP/s
  --mode sharded ONE GP of size --n sharded over the ranks (column-panel block-cyclic Cholesky, RCCL broadcast of
                 factored panels, battgp_amd/sharded.py; BASELINE configs[3]); strong scaling
With --mode cells and N > 1 a `sharded` sub-record (one GP of --sharded-n points over all ranks) is appended so that a
scaling run measures both multi-GPU paths.  Prints ONE JSON line on rank 0.
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
T_PROCESS_START = time.perf_counter()

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak (= 256 CU x 2.4 GHz x 128 flop/clk)
PEAK_HBM_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PROFILE_TAG = "r06"  # profiles/<tag>_n<N>_<kernel>_summary.json: committed rocprofv3 PMC passes of this round


# ---------------------------------------------------------------------------------------------------------------
# The record is durable: once the timed region is over, whatever ends the process (the caller's time limit = SIGTERM,
# Ctrl-C) prints the ONE JSON line with what has been measured so far; side measurements run in their own sessions
# so that a time limit (theirs or ours) ends the whole child tree - a profiler's grandchild would otherwise keep
# its N^2 factor in HBM underneath the next pass.
# ---------------------------------------------------------------------------------------------------------------
_STATE = {"out": None, "printed": False, "children": []}


def emit(out) -> None:
    if out is not None and not _STATE["printed"]:
        _STATE["printed"] = True
        print(json.dumps(out), flush=True)


def run_child(cmd, timeout, **kw):
    """subprocess.run(capture_output=True, text=True) in a session of its own; on a timeout the whole session is killed
    (by the process-group id created here) before subprocess.TimeoutExpired is re-raised."""
    import signal
    import subprocess

    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    _STATE["children"].append(p)
    try:
        so, se = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
        p.communicate()
        raise
    finally:
        _STATE["children"].remove(p)
    return subprocess.CompletedProcess(cmd, p.returncode, so, se)


def _on_term(signum, _frame):
    import signal

    for p in list(_STATE["children"]):
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    out = _STATE["out"]
    if out is not None and not _STATE["printed"]:
        out["interrupted"] = f"signal {signum} during the side measurements: the record holds what was finished by then"
        out.setdefault("cpu_baseline", None)
        emit(out)
    sys.stdout.flush()
    os._exit(0 if out is not None else 128 + signum)


def algorithmic_flop(n: int, m: int) -> float:
    """N^3/3 (Cholesky) + N^2 M (predictive TRSM) + 2 N^2 (two TRSV)."""
    return n**3 / 3.0 + float(n) * n * m + 2.0 * n * n


def vs_n_need_s(n: int, m: int) -> float:
    """side budget one point of the "vs N" curve must find left: four steps at 40 TFLOP/s + 3 s (allocation, data)"""
    return 4.0 * algorithmic_flop(n, m) / 40e12 + 3.0


def kernel_setup(name: str):
    from battgp_amd import KERNEL_BATTGP, KERNEL_MATERN32, synthetic

    if name == "battgp":
        return KERNEL_BATTGP, synthetic.HYP_BATTGP, "Wiener+ARD-RBF (reference full_gp kernel)"
    return KERNEL_MATERN32, synthetic.HYP_MATERN32, "Matern-3/2 ARD + noise"


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = port of the reference algorithm; checker code timed as a baseline, never the product)
# ---------------------------------------------------------------------------------------------------------------
OPENBLAS_POTRF_LIMIT = 32768  # scipy's / numpy's OpenBLAS dpotrf of this image segfaults from N = 2^15 on (seen at 32 700 / 32 800;
                              # 30 000 is fine; torch's MKL is unaffected) - never call it there


def cpu_baseline(kernel_name: str, n_cpu: int, m: int, n_second: int = 0, budget_s: float = 150.0, progress=None) -> dict:
    """The oracle's dense algorithm on the host cores with its phases timed separately: threaded numpy fill, LAPACK
    dpotrf, triangular solves + posterior.  The factorisation is timed with BOTH LAPACKs of this image - scipy's
    OpenBLAS (built for at most 64 threads, whatever the host has) and torch's CPU LAPACK (MKL, `torch.get_num_threads()`
    threads, no such cap) - and the faster one counts: `value` = algorithmic flop / (fill + best potrf + solves),
    `cores` = the threads of that LAPACK.  `n_second` > 0 adds a second sample at that size (BASELINE configs[1]'s
    N = 40 000) when the first sample's rate says it fits `budget_s`."""
    import scipy.linalg as sla
    from threadpoolctl import threadpool_info

    from battgp_amd import synthetic
    from oracle import kernels as K

    kid, hyp, _ = kernel_setup(kernel_name)
    blas_threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    try:
        import torch

        torch_threads = int(torch.get_num_threads())
    except Exception:  # noqa: BLE001
        torch, torch_threads = None, 0

    def run(n, both=True):
        x, y = synthetic.make_cell_data(n, seed=n)
        xq = synthetic.make_query(x, m)
        t0 = time.perf_counter()
        sigma = K.kernel_matrix_blocked(kid, hyp, x) if n > 2048 else K.kernel_matrix(kid, hyp, x)
        sigma[np.diag_indices(n)] += K.noise(hyp)
        t1 = time.perf_counter()
        potrf = {}
        chol = None
        big = n >= OPENBLAS_POTRF_LIMIT - 1024  # (margin: the library's limit was not bisected below 32 700)
        if torch is not None and (both or big):
            try:
                tt = time.perf_counter()
                lt = torch.linalg.cholesky(torch.from_numpy(sigma))
                potrf["torch_cpu_lapack"] = {"seconds": time.perf_counter() - tt, "threads": torch_threads}
                if big:
                    chol = lt.numpy()  # this factor feeds the solves: OpenBLAS must not see a matrix of this size
                del lt
            except Exception as exc:  # noqa: BLE001 - the scipy path below is the one the solves use
                potrf["torch_cpu_lapack"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        if not big:
            tt = time.perf_counter()
            chol, info = sla.lapack.dpotrf(sigma, lower=1, clean=0, overwrite_a=1)
            assert info == 0
            potrf["scipy_openblas"] = {"seconds": time.perf_counter() - tt, "threads": int(blas_threads)}
        else:
            potrf["scipy_openblas"] = {"skipped": f"OpenBLAS dpotrf of this image segfaults for N >= {OPENBLAS_POTRF_LIMIT}; torch's LAPACK factors this sample"}
            if chol is None:
                raise RuntimeError("no LAPACK available for a sample of this size (torch missing or its factorisation failed)")
        t2 = time.perf_counter()
        kxs = K.kernel_matrix(kid, hyp, x, xq)
        if big:
            # the same OpenBLAS also dies in level-3 products from N = 32 768 on (dtrsm here): the solves of a big sample
            # go through torch's LAPACK like its factorisation
            lt = torch.from_numpy(chol)
            rhs = torch.from_numpy(np.ascontiguousarray(np.column_stack([y, kxs])))
            sol = torch.linalg.solve_triangular(lt, rhs, upper=False).numpy()
            z, v = sol[:, 0].copy(), sol[:, 1:]
            del lt, rhs
        else:
            z = sla.solve_triangular(chol, y, lower=True, check_finite=False)
            v = sla.solve_triangular(chol, kxs, lower=True, check_finite=False)
        mean = v.T @ z
        var = K.kernel_diag(kid, hyp, xq) - np.einsum("ij,ij->j", v, v)
        lml = -0.5 * float(z @ z) - float(np.sum(np.log(np.diag(chol)))) - 0.5 * n * np.log(2 * np.pi)
        t3 = time.perf_counter()
        best = min((k for k in potrf if "seconds" in potrf[k]), key=lambda k: potrf[k]["seconds"])
        potrf_s = potrf[best]["seconds"]
        total = (t1 - t0) + potrf_s + (t3 - t2)
        for k in potrf:
            if "seconds" in potrf[k]:
                potrf[k]["gflops"] = (n**3 / 3.0) / potrf[k]["seconds"] / 1e9
        return {
            "n": n, "seconds": total, "fill_s": t1 - t0, "potrf_s": potrf_s, "solve_predict_s": t3 - t2,
            "gflops": algorithmic_flop(n, m) / total / 1e9, "potrf_gflops": (n**3 / 3.0) / potrf_s / 1e9,
            "potrf_by_lapack": potrf, "potrf_lapack_used": best, "threads": potrf[best]["threads"],
            "lml": lml, "mean0": float(mean[0]), "var0": float(var[0]),
        }

    main = run(n_cpu)
    cfg0 = run(2048)  # BASELINE configs[0]: the reference's own CPU-runnable case
    rec = {
        "value": main["gflops"],
        "unit": "GFLOP/s",
        "cores": int(main["threads"]),
        "host_cpus": os.cpu_count(),
        "fill_threads": min(32, os.cpu_count() or 1),
        "kind": "port",
        "seconds": main["seconds"],
        "phases": {k: main[k] for k in ("fill_s", "potrf_s", "solve_predict_s")},
        "potrf_gflops_lapack_alone": main["potrf_gflops"],
        "potrf_by_lapack": main["potrf_by_lapack"],
        "config0_n2048": cfg0,
        "sample": f"same workload ({kernel_name}) at N={n_cpu}, one fit+predict with M={m}: oracle algorithm = numpy fill on "
                  f"{min(32, os.cpu_count() or 1)} threads + LAPACK dpotrf/dtrtrs; dpotrf timed with scipy/OpenBLAS ({blas_threads} threads: "
                  f"its build-time cap, host has {os.cpu_count()} CPUs) and torch CPU LAPACK ({torch_threads} threads), faster one counted "
                  f"({main['potrf_lapack_used']})",
    }
    if progress is not None:
        progress(rec)  # the first sample is safe whatever the second one does
    if n_second > n_cpu:
        est = main["seconds"] * (n_second / n_cpu) ** 3 * (2.0 if len(main["potrf_by_lapack"]) > 1 else 1.0)
        if est <= budget_s:
            try:
                rec["second_sample"] = run(n_second)
            except Exception as exc:  # noqa: BLE001
                rec["second_sample"] = {"n": n_second, "error": f"{type(exc).__name__}: {exc}"[:200]}
        else:
            rec["second_sample"] = {"n": n_second, "skipped": f"estimated {est:.0f} s from the N={n_cpu} rate, budget {budget_s:.0f} s"}
        if progress is not None:
            progress(rec)
    return rec


# ---------------------------------------------------------------------------------------------------------------
# PMC evidence: collected live by child rocprofv3 passes of this workload (pmc_live), else the committed passes
# ---------------------------------------------------------------------------------------------------------------
TRAIL_FRAG = "gemm_nt_kernel<128, 128, 2"  # the rank-NB trailing update in rocprofv3's kernel names


def pmc_live(n: int, kernel: str, m: int = 300, limit_s: float = 240.0) -> dict:
    """HBM traffic and MFMA utilisation of the trailing update at THIS run's workload: rocprofv3 --pmc passes over
    tools/profile_workload.py (one fused fit+predict + alpha()) in child processes, one counter group per pass
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domain next to --pmc), each with a time limit.
    Corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are KiB; on gfx950 FETCH_SIZE tallies a
    wide coalesced read at half its bytes (x2).  Calibrated inside the same passes on two known byte counts:
    gemv_t_partial reads the strictly-lower panels of L exactly once, the fill writes 4N(N+1) + 8MN bytes once."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = os.environ.get("BGP_ROCPROFV3") or shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"rc": "rocprofv3 not found"}
    work = tempfile.mkdtemp(prefix="bgp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))  # kernel -> counter -> [dispatches, sum]
    span = collections.defaultdict(float)  # kernel -> summed dispatch duration (ns) of the MFMA pass
    passes = {}
    try:
        for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                              ("mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])):
            d = os.path.join(work, tag)
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "profile_workload.py"), str(n), kernel]
            t0 = time.perf_counter()
            try:
                r = run_child(cmd, limit_s, cwd="/tmp", env=env)
                passes[tag] = {"rc": r.returncode, "s": time.perf_counter() - t0}
                if r.returncode:
                    passes[tag]["stderr_tail"] = r.stderr[-300:]
            except subprocess.TimeoutExpired:
                passes[tag] = {"rc": "timeout", "s": limit_s}
                continue
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        a = agg[row["Kernel_Name"]][row["Counter_Name"]]
                        a[0] += 1
                        a[1] += float(row["Counter_Value"])
                        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                            span[row["Kernel_Name"]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    finally:
        shutil.rmtree(work, ignore_errors=True)

    def pick(frag, counter):
        ks = [k for k in agg if frag in k and counter in agg[k]]
        return sum(agg[k][counter][0] for k in ks), sum(agg[k][counter][1] for k in ks)

    kib, xccs, simds = 1024.0, 8, 1024
    npad = (n + 63) // 64 * 64
    nb = 1024 if npad >= 32768 else 512  # the engine's automatic outer panel width (apply_auto_nb)
    calls_f, fetch = pick(TRAIL_FRAG, "FETCH_SIZE")
    calls_w, write = pick(TRAIL_FRAG, "WRITE_SIZE")
    _, busy = pick(TRAIL_FRAG, "SQ_VALU_MFMA_BUSY_CYCLES")
    _, gui = pick(TRAIL_FRAG, "GRBM_GUI_ACTIVE")
    dur_ns = sum(v for k, v in span.items() if TRAIL_FRAG in k)
    gemv_expected = sum(8.0 * (npad - min(k0 + nb, npad)) * (min(k0 + nb, npad) - k0) for k0 in range(0, npad, nb))
    _, gemv_fetch = pick("gemv_t_partial_kernel", "FETCH_SIZE")
    _, fill_write = pick("fill_kernel", "WRITE_SIZE")
    out = {"passes": passes, "workload": f"tools/profile_workload.py {n} {kernel}: one fused fit+predict (M = {m}) + alpha()"}
    if calls_f and calls_w:
        out.update({
            "dispatches": calls_f,
            "fetch_bytes_corrected_per_dispatch": fetch * kib * 2.0 / calls_f,
            "write_bytes_per_dispatch": write * kib / calls_w,
            "hbm_bytes_per_dispatch": fetch * kib * 2.0 / calls_f + write * kib / calls_w,
            "hbm_bytes_total": fetch * kib * 2.0 + write * kib,
            # every trailing element read and written once per outer panel (the atomic-add epilogue): 2 * 8 * N^3 / (3 NB) / 2
            "algorithmic_c_traffic_bytes_total": 2 * 4.0 * n**3 / (3 * nb),
            "calibration": {
                "fetch_raw_over_expected_gemv_t (0.5 = the guide's x2 correction holds)": gemv_fetch * kib / gemv_expected if gemv_expected else None,
                "fill_write_over_algorithmic": fill_write * kib / (4.0 * n * (n + 1) + 8.0 * m * n),
            },
        })
    if gui:
        out["mfma_util"] = busy / (gui / xccs * simds)
        out["effective_clock_ghz_under_pmc"] = (gui / xccs) / dur_ns if dur_ns else None
    return out


def profile_summary(n: int, kernel: str):
    for tag in (PROFILE_TAG, "r01"):
        for name in (f"{tag}_n{n}_{kernel}_summary.json", f"{tag}_n{n}_summary.json" if kernel == "battgp" else None):
            if name and os.path.exists(os.path.join(ROOT, "profiles", name)):
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    return json.load(f), name
    return None, None


def n_max_need_bytes(n: int, m: int, slab: int, nb: int) -> float:
    """Device memory of ONE fit+predict in the column-slab layout (bgp_capi.hip alloc_problem / choose_slab_width):
    ~4 lda (Npad + W) bytes of factor, two solved-panel buffers of the panel scheme, inputs and solve vectors, runtime.
    Held against profiles/r01_large_n.json: N = 274 432, W = 2048, nb = 512 -> 306.76 GB measured, 306.80 here;
    N = 262 144, W = 16 384, nb = 512 -> 293.01 GB measured, 295.26 here (wide slabs: an over-estimate, the safe side)."""
    npad = (n + 63) // 64 * 64
    lda = npad + 64 + (m + 63) // 64 * 64
    w = slab if slab > 0 else 16384
    return 4.0 * lda * (npad + w) + 2.0 * lda * nb * 8.0 + 80.0 * n + 0.6e9


def n_max_size_for(free_b: float, n_req: int, m: int, slab: int, nb: int, margin_b: float = 3e9, floor_n: int = 196608) -> int:
    """Largest N <= n_req (steps of 4096 - a whole number of panels at every width in use) whose fit leaves `margin_b` of the
    HBM that is free NOW: the parent process keeps its HIP context while the child runs (at N = 274 432 the
    child needs 306.76 of the 308.56 GB an empty GPU has).  0 when not even `floor_n` (the full-square ceiling) fits."""
    n = int(n_req)
    floor_n = min(floor_n, n)  # (a request below the ceiling - the rehearsals - is taken as it is or not at all)
    while n >= floor_n:
        if n_max_need_bytes(n, m, slab, nb) + margin_b <= free_b:
            return n
        n -= 4096
    return 0


def n_max_measured(n_max: int, kernel: str, m: int, slab: int, nb: int, limit_s: float) -> dict:
    """BASELINE's "N_max per GPU", measured by THIS run: one cold fit+predict at the largest size one MI355X holds (the factor
    in column slabs, bgp_set_layout) - tools/large_n.py in a child process with a time limit (its own session: a device
    allocation that does not fit, or a run-away, costs this field only).  ~105 s at N = 274 432."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "large_n.py"), str(n_max), kernel, str(slab), str(m), str(nb), "1"]
    t0 = time.perf_counter()
    r = run_child(cmd, limit_s)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"tools/large_n.py {n_max} ended with rc {r.returncode}: {r.stderr[-250:]}")
    rec = json.loads(lines[-1])
    return {
        "n": rec["n"], "fit_predict_s": rec["fit_predict_s"], "gflops": rec["gflops"], "potrf_tflops": rec["potrf_tflops"],
        "frac_of_peak": rec["gflops"] / 1e3 / PEAK_FP64_MFMA_TFLOPS, "slab_width": rec["slab_width"], "nb_outer": rec["nb_outer"],
        "factor_bytes": rec["factor_bytes"], "device_bytes": rec["device_bytes"], "hbm_total": rec["hbm_total"],
        "residuals": rec["residuals"], "lml": rec["lml"], "jitter": rec["jitter"], "full_square_limit_n": 196000, "kernel": kernel,
        "child_s": time.perf_counter() - t0,
        "source": "measured by this run: ONE cold call of tools/large_n.py (includes the hipMalloc of the factor) in a child process",
    }


def n_max_from_profile():
    """FALLBACK of n_max_measured (side budget spent, --no-extras, N > 1): the committed record of the largest N measured on
    one MI355X (column-slab layout of the factor, tools/large_n.py) - a citation, labelled as one."""
    path = os.path.join(ROOT, "profiles", "r01_large_n.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        runs = json.load(f)["runs"]
    best = max(runs, key=lambda r: (r["n"], r.get("panel_scheme", 0)))
    return {
        "n": best["n"], "fit_predict_s": best["fit_predict_s"], "gflops": best["gflops"], "slab_width": best["slab_width"],
        "factor_bytes": best["factor_bytes"], "residuals": best["residuals"], "full_square_limit_n": 196000,
        "source": "CITED, not measured by this run: profiles/r01_large_n.json (tools/large_n.py, round 1)",
    }


# ---------------------------------------------------------------------------------------------------------------
# one workload on this rank's GPU
# ---------------------------------------------------------------------------------------------------------------
class CellWorkload:
    def __init__(self, n, m, kernel, device_index, seed, args):
        import torch

        from battgp_amd import synthetic
        from battgp_amd.engine import ExactGPEngine

        self.torch, self.n, self.m = torch, n, m
        kid, hyp, _ = kernel_setup(kernel)
        x, y = synthetic.make_cell_data(n, seed=seed)
        xq = synthetic.make_query(x, m)
        dev = torch.device("cuda", device_index)
        self.tx, self.ty, self.txq = (torch.from_numpy(a).to(dev) for a in (x, y, xq))
        self.tmean = torch.empty(m, dtype=torch.float64, device=dev)
        self.tvar = torch.empty(m, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        self.eng = ExactGPEngine(kid, hyp, device=device_index)
        if args.nb > 0:
            self.eng.set_options(nb_outer=args.nb)
        if args.lookahead >= 0:
            self.eng.set_options(lookahead=args.lookahead)
        if args.panel_scheme >= 0:
            self.eng.set_panel_scheme(args.panel_scheme)
        if args.slab != 0:
            self.eng.set_layout(args.slab)
        self.separate = args.separate

    def step(self):
        e = self.eng
        if self.separate:
            e.fit_device(self.tx.data_ptr(), self.ty.data_ptr(), self.n, 4)
            e.predict_device(self.txq.data_ptr(), self.m, self.tmean.data_ptr(), self.tvar.data_ptr(), 1e-10)
        else:  # the reference's flow: the first predict triggers the factorisation; one fused pass
            e.fit_predict_device(self.tx.data_ptr(), self.ty.data_ptr(), self.n, 4, self.txq.data_ptr(), self.m,
                                 self.tmean.data_ptr(), self.tvar.data_ptr(), 1e-10)

    def fill_steady_gbs(self, reps=8):
        """The fill kernel launched back to back on the factor's own footprint-sized scratch (its steady rate: no
        clock ramp after a host-side gap - tools/fill_gap_probe.py)."""
        torch, n = self.torch, self.n
        free_b, _ = torch.cuda.mem_get_info()
        ld = n + 384
        if free_b < ld * n * 8 + (4 << 30):
            return None
        scratch = torch.empty((n, ld), dtype=torch.float64, device=self.tx.device)
        torch.cuda.synchronize()
        rates = []
        for _ in range(reps):
            self.eng.fill_device(self.tx.data_ptr(), n, self.tx.data_ptr(), n, 4, scratch.data_ptr(), ld, lower=1, diag_add=float(self.eng.hyp[0]))
            ph = self.eng.phase_times()
            rates.append(ph["fill_bytes"] / (ph["fill_ms"] * 1e-3) / 1e9)
        del scratch
        torch.cuda.empty_cache()
        return float(np.median(rates[reps // 2:]))

    def memset_gbs(self, reps=4):
        """hipMemset of the same number of bytes: the write-stream ceiling of this GPU as a library call sees it."""
        torch, n = self.torch, self.n
        nbytes = 4 * n * (n + 1)
        free_b, _ = torch.cuda.mem_get_info()
        if free_b < nbytes + (4 << 30):
            return None
        buf = torch.empty(nbytes // 8, dtype=torch.float64, device=self.tx.device)
        torch.cuda.synchronize()
        rates = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            buf.zero_()
            b.record()
            b.synchronize()
            rates.append(nbytes / (a.elapsed_time(b) * 1e-3) / 1e9)
        del buf
        torch.cuda.empty_cache()
        return float(np.median(rates[1:]))

    def close(self):
        self.eng.close()


def summarise(phases, n, m, kernel, desc):
    """Roofline records from the HIP-event phase timers of the timed steps (events on the launching streams,
    inside the library)."""
    avg = {k: float(np.mean([p[k] for p in phases])) for k in phases[0]}
    trail_tflops = avg["trail_flop"] / (avg["trail_ms"] * 1e-3) / 1e12 if avg["trail_ms"] > 0 else 0.0
    union = avg.get("trail_union_ms", 0.0)
    fill_gbs = avg["fill_bytes"] / (avg["fill_ms"] * 1e-3) / 1e9 if avg["fill_ms"] > 0 else 0.0
    potrf_tflops = (n**3 / 3.0) / (avg["potrf_ms"] * 1e-3) / 1e12 if avg["potrf_ms"] > 0 else 0.0
    prof, prof_name = profile_summary(n, kernel)
    gem = (prof or {}).get("gemm_nt_128x128", {})
    launches = int(round(avg["trail_launches"]))
    roofline = {
        "bound": "mfma",
        "kernel": "gemm_nt_kernel<128,128,2> (rank-NB SYRK trailing update of the blocked Cholesky)",
        "achieved": trail_tflops,
        "peak": PEAK_FP64_MFMA_TFLOPS,
        "unit": "TFLOP/s",
        "frac": trail_tflops / PEAK_FP64_MFMA_TFLOPS,
        "launches_per_step": launches,
        "flop_per_launch": avg["trail_flop"] / max(1, launches),
        "avg_launch_ms": avg["trail_ms"] / max(1, launches),
        "achieved_while_running": (avg["trail_flop"] / (union * 1e-3) / 1e12) if union > 0 else None,
        "frac_while_running": (avg["trail_flop"] / (union * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if union > 0 else None,
        "traffic": gem.get("hbm_bytes_per_dispatch"),
        "traffic_source": f"profiles/{prof_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; not re-collected by this run)" if gem else None,
        "mfma_util_pmc": gem.get("mfma_util"),
        "note": "achieved = sum of algorithmic flop 2k*#{i>=j} of the outer trailing-update launches / sum of their HIP-event "
                "durations (= flop per launch / average launch duration); with look-ahead the la(k) and rest(k) launches overlap "
                "on two streams, so achieved_while_running divides the same flop by the UNION of the launch intervals",
    }
    roofline_fill = {
        "bound": "hbm",
        "kernel": "fill_kernel (lower triangle, 4N(N+1) algorithmic bytes per fit)",
        "achieved": fill_gbs,
        "peak": PEAK_HBM_GBS,
        "unit": "GB/s",
        "frac": fill_gbs / PEAK_HBM_GBS,
        "fill_ms": avg["fill_ms"],
        "traffic": (prof or {}).get("fill_kernel", {}).get("hbm_bytes_per_dispatch"),
        "note": "in situ: the fill launches of the timed steps, HIP events around them inside the fit",
    }
    return {
        "workload": f"full_gp, {desc}, N={n} synthetic 4-D inputs, M={m} queries",
        "roofline": roofline,
        "roofline_fill": roofline_fill,
        "phases_ms": {k: avg[k] for k in ("h2d_ms", "fill_ms", "potrf_ms", "solve_ms", "cross_ms", "var_ms", "d2h_ms", "trail_ms")},
        "potrf_tflops": potrf_tflops,
        "potrf_frac_of_peak": potrf_tflops / PEAK_FP64_MFMA_TFLOPS,
    }


def run_cells(args, rank, world, local_rank, dist, red_dev):
    import torch

    from battgp_amd import parallel

    n, m = args.n, args.m
    _, _, desc = kernel_setup(args.kernel)
    # every rank = a different cell (different seed), same size: weak scaling, no collective
    wl = CellWorkload(n, m, args.kernel, local_rank, n + rank, args)

    def barrier():
        parallel.barrier(dist)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    phases = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()  # every C-ABI call returns only after its own stream has drained
        phases.append(wl.eng.phase_times())
    barrier()
    elapsed = parallel.max_over_ranks(dist, time.perf_counter() - t0, device=red_dev)

    resid = None if args.no_residuals else wl.eng.residuals(256)  # on-device evidence at the benchmarked size
    mean_host = wl.tmean.cpu().numpy()
    lml, jitter, mem_bytes = wl.eng.lml, wl.eng.jitter, wl.eng.device_bytes()
    out = None
    if rank == 0:
        flop = algorithmic_flop(n, m)
        rec = summarise(phases, n, m, args.kernel, desc)
        out = {
            "metric": "exact-GP fit+predict throughput (fp64)",
            "value": world * flop * args.steps / elapsed / 1e9,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": rec["workload"] + ", one cell per GPU",
                "n": n, "m": m, "kernel": args.kernel, "flop_per_step": flop,
                "parallelism": f"{world} independent cells" if world > 1 else "1 cell",
            },
            "roofline": rec["roofline"],
            "roofline_fill": rec["roofline_fill"],
            "phases_ms": rec["phases_ms"],
            "potrf_tflops": rec["potrf_tflops"],
            "potrf_frac_of_peak": rec["potrf_frac_of_peak"],
            "lml": lml,
            "jitter": jitter,
            "residuals": {"rel_solve": resid[0], "max_llt": resid[1]} if resid else None,
            "mean_first": [float(v) for v in mean_host[:3]],
            "device_bytes": mem_bytes,
            "n_max_per_gpu": n_max_from_profile(),
        }
    wl.eng.close()  # frees (or parks) the factor before the scratch-sized side measurements
    _STATE["out"] = out  # from here on a SIGTERM prints the record (rank 0) instead of losing it
    if out is not None:
        # ... and a SIGKILL (a caller's hard time limit) during the side measurements still leaves the timed region on
        # STDERR (stdout carries exactly one JSON line, at the end)
        print("bench.py: timed region done, side measurements follow; record so far: " + json.dumps(out), file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_extras:
        from battgp_amd.engine import trim_pool

        side_errors = {}

        t_side0 = time.perf_counter()
        # the side measurements get --side-budget-s, but never more than what --wall-budget-s leaves of the WHOLE run: with the
        # driver's --steps 20 --warmup 5 the timed region alone takes ~4.5 min at the headline size, and a caller's time limit
        # that kills outright (SIGKILL: no chance to print) would cost the record
        side_budget = min(args.side_budget_s, max(0.0, args.wall_budget_s - (t_side0 - T_PROCESS_START)))
        out["side_budget_s"] = side_budget

        def remaining():
            return side_budget - (time.perf_counter() - t_side0)

        def side(name, fn, need_s=0.0):
            """A side measurement never takes the record above down with it, and none is started once the time budget
            for all of them (--side-budget-s) can no longer cover it."""
            if remaining() < need_s:
                side_errors[name] = f"skipped: {remaining():.0f} s of the side budget left, needs ~{need_s:.0f} s"
                return None
            try:
                return fn()
            except Exception as exc:  # noqa: BLE001
                side_errors[name] = f"{type(exc).__name__}: {exc}"[:300]
                return None

        def fill_side():
            # steady rate of the fill kernel and the memset ceiling, beside the in-situ figure
            wl2 = CellWorkload(n, m, args.kernel, local_rank, n, args)
            trim_pool(local_rank)
            try:
                out["roofline_fill"]["steady_gbs"] = wl2.fill_steady_gbs()
                out["roofline_fill"]["steady_frac"] = (out["roofline_fill"]["steady_gbs"] or 0.0) / PEAK_HBM_GBS
                out["roofline_fill"]["hipmemset_same_bytes_gbs"] = wl2.memset_gbs()
            finally:
                wl2.close()
                trim_pool(local_rank)

        def extra_config(en, ek):
            w = CellWorkload(en, m, ek, local_rank, en, args)
            try:
                w.step()
                ph, t0 = [], time.perf_counter()
                for _ in range(3):
                    w.step()
                    ph.append(w.eng.phase_times())
                dt = (time.perf_counter() - t0) / 3
                rec = summarise(ph, en, m, ek, kernel_setup(ek)[2])
                res = w.eng.residuals(256)
                rec.update({"ms_per_step": dt * 1e3, "gflops": algorithmic_flop(en, m) / dt / 1e9, "lml": w.eng.lml,
                            "residuals": {"rel_solve": res[0], "max_llt": res[1]}})
                return rec
            finally:
                w.close()

        def sweep_point(sn):
            """one point of BASELINE's "vs N" curve: median of warm fused fit+predict calls, inputs resident in HBM"""
            w = CellWorkload(sn, m, args.kernel, local_rank, sn, args)
            try:
                w.step()
                ts = []
                for _ in range(5 if sn <= 16384 else 3):
                    t0 = time.perf_counter()
                    w.step()
                    ts.append(time.perf_counter() - t0)
                dt, ph = float(np.median(ts)), w.eng.phase_times()
                gf = algorithmic_flop(sn, m) / dt / 1e9
                return {"n": sn, "ms": dt * 1e3, "gflops": gf, "frac_of_peak": gf / 1e3 / PEAK_FP64_MFMA_TFLOPS, "reps": len(ts),
                        "potrf_ms": ph["potrf_ms"], "fill_ms": ph["fill_ms"], "lml": w.eng.lml, "jitter": w.eng.jitter}
            finally:
                w.close()

        # REQUIRED field before optional evidence: `cpu_baseline` (host cores only, a child process) is the first thing after
        # the timed region - a caller's time limit that ends the side measurements early then costs optional fields only.
        # Three profiler passes of one fit+predict each (+ process start): the step time is the best estimate of a pass.
        pass_s = 3.0 * (elapsed / args.steps) + 30.0
        # The first CPU sample is required; the second (N = 40 000) is optional and only gets what the side budget has left
        # once the profiler passes and the GPU-side extras (steady fill, extra config, vs-N curve: ~60 s) are paid for
        host_baseline(out, args, second_budget_s=min(args.cpu_budget_s, remaining() - (0.0 if args.no_pmc else 3.0 * pass_s) - 90.0),
                      limit_s=max(120.0, remaining() - 10.0))
        side("fill_steady", fill_side, 20.0)
        extras = [side(f"extra_{en}_{ek}", lambda en=en, ek=ek: extra_config(en, ek), 10.0)
                  for en, ek in ((args.extra_n, "battgp"),) if en > 0 and not (en == n and ek == args.kernel)]
        out["extra_configs"] = [e for e in extras if e]
        # BASELINE's metric is "... vs N": the curve below the headline size, the headline point from the timed region itself,
        # N_max appended further down when it was measured.  A point = one cold call (hipMalloc of the factor) + 3-5 warm
        # steps: ~4 steps at >= 40 TFLOP/s + 3 s of set-up (N = 65 536: ~12 s; the five default points ~20 s in all)
        pts = [side(f"vs_n_{sn}", lambda sn=sn: sweep_point(sn), vs_n_need_s(sn, m)) for sn in args.sweep_n if 0 < sn < n]
        gf0 = flop * args.steps / elapsed / 1e9
        out["vs_n"] = [p for p in pts if p] + [{"n": n, "ms": elapsed / args.steps * 1e3, "gflops": gf0, "frac_of_peak": gf0 / 1e3 / PEAK_FP64_MFMA_TFLOPS,
                                                 "reps": args.steps, "potrf_ms": rec["phases_ms"]["potrf_ms"], "fill_ms": rec["phases_ms"]["fill_ms"],
                                                 "lml": lml, "jitter": jitter, "note": "the timed region of this run"}]
        trim_pool(local_rank)  # the children below need the HBM
        if not args.no_pmc:
            live = side("pmc_live", lambda: pmc_live(n, args.kernel, m, limit_s=max(60.0, min(180.0, remaining() / 3.0))), 3.0 * pass_s)
            out["pmc_live"] = live
            if live and live.get("hbm_bytes_per_dispatch"):
                out["roofline"]["traffic"] = live["hbm_bytes_per_dispatch"]
                out["roofline"]["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload run by this "
                                                     "bench in child processes (pmc_live); FETCH_SIZE x2 per the gfx950 correction")
                out["roofline"]["traffic_over_algorithmic_c_traffic"] = live["hbm_bytes_total"] / live["algorithmic_c_traffic_bytes_total"]
            if live and live.get("mfma_util") is not None:
                out["roofline"]["mfma_util_pmc"] = live["mfma_util"]
        if args.nmax_n > 0:
            # the last side measurement (the longest, ~105 s + process start): only with >= 150 s of the budget left.
            # The parent lets go of what it holds first (the engines are closed; their parked buffers and torch's cached
            # blocks are not), then the size is chosen from the HBM that is really free next to the parent's own context
            trim_pool(local_rank)
            free_b = None
            try:
                import torch

                torch.cuda.empty_cache()
                free_b = float(torch.cuda.mem_get_info(local_rank)[0])
            except Exception as exc:  # noqa: BLE001 - the citation below stays
                side_errors["n_max_free_hbm"] = f"{type(exc).__name__}: {exc}"[:200]
            nmax_n = n_max_size_for(free_b, args.nmax_n, m, args.nmax_slab, args.nmax_nb) if free_b is not None else args.nmax_n
            if nmax_n <= 0:
                side_errors["n_max"] = f"skipped: {free_b / 1e9:.1f} GB of HBM free next to this process, not enough for N >= 196608 in slabs"
                got = None
            else:
                got = side("n_max", lambda: n_max_measured(nmax_n, args.kernel, m, args.nmax_slab, args.nmax_nb, max(30.0, remaining())), args.nmax_need_s)
            if got:
                got["hbm_free_before"], got["n_requested"] = free_b, args.nmax_n
                out["n_max_per_gpu"] = got
                out["vs_n"].append({"n": got["n"], "ms": got["fit_predict_s"] * 1e3, "gflops": got["gflops"], "frac_of_peak": got["frac_of_peak"],
                                    "reps": 1, "lml": got["lml"], "jitter": got["jitter"], "note": "N_max: one COLD call, column-slab layout"})
            elif out.get("n_max_per_gpu"):
                out["n_max_per_gpu"]["fallback_because"] = side_errors.get("n_max", "not attempted")
                out["n_max_per_gpu"]["hbm_free_before"], out["n_max_per_gpu"]["n_attempted"] = free_b, nmax_n
        # the A/B of the never-measured optional schedules is an experiment of a builder's session (tools/gpu_session.sh), not
        # part of the default record: a kernel that hangs the GPU cannot be recovered by killing its child process
        out["experiments"] = side("experiments", lambda: schedule_experiments(max(30.0, min(150.0, remaining()))), 30.0) if args.experiments else None
        if side_errors:
            out["side_measurement_errors"] = side_errors
    return out


def host_baseline(out, args, second_budget_s=None, limit_s=None) -> None:
    """out["cpu_baseline"]: the oracle's dense path on this box's host cores (bounded samples), measured in a CHILD process
    (`bench.py --mode cpu_baseline`) that prints its record after every sample: a LAPACK that crashes on a large matrix
    (this image's OpenBLAS dpotrf does from N = 32 768 on) or runs away costs at most the later sample, never the GPU
    record.  Never raises."""
    if "cpu_baseline" in out or args.cpu_n <= 0:
        return
    budget = args.cpu_budget_s if second_budget_s is None else max(0.0, float(second_budget_s))
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "cpu_baseline", "--kernel", args.kernel, "--m", str(args.m),
           "--cpu-n", str(args.cpu_n), "--cpu-n2", str(args.cpu_n2 if budget > 0 else 0), "--cpu-budget-s", str(budget)]
    rec, note = None, ""
    try:
        # the child prints its record after every finished sample: a limit that cuts the second sample off keeps the first.
        # `limit_s` = what the caller's own time budget leaves (never below the ~2 min the first, required sample may take)
        r = run_child(cmd, budget + 240.0 if limit_s is None else min(budget + 240.0, float(limit_s)))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"cpu_baseline"')]
        if lines:
            rec = json.loads(lines[-1])["cpu_baseline"]
        if r.returncode != 0:
            note = f"child ended with rc {r.returncode} after {len(lines)} record(s): {r.stderr[-200:]}"
    except Exception as exc:  # noqa: BLE001 - the GPU record is printed whatever happens to the host-side sample
        note = f"{type(exc).__name__}: {exc}"[:300]
        so = getattr(exc, "stdout", None) or ""
        lines = [ln for ln in (so if isinstance(so, str) else so.decode(errors="replace")).splitlines() if ln.startswith('{"cpu_baseline"')]
        if lines:
            rec = json.loads(lines[-1])["cpu_baseline"]
    if rec is None:
        rec = {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "port", "sample": f"failed: {note}"[:300]}
    elif note:
        rec["child_note"] = note
    out["cpu_baseline"] = rec


def schedule_experiments(limit_s: float = 150.0):
    """A/B of the optional Cholesky schedules (tools/ab_lookahead.py) in a CHILD process with a time limit: they are
    off by default until measured, and nothing they do can reach the record above."""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "tools", "ab_lookahead.py"), "3", "4096", "16384", "40000"]
    try:
        # the optional schedules exist in the experimental library only (battgp_amd/build.py --experimental)
        r = run_child(cmd, limit_s, env=dict(os.environ, BGP_EXPERIMENTAL_LIB="1"))
        lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        return {"what": "fit+predict ms by lookahead word (1 default, +32 slim chain kernels, +64 split panels, +128 fused update + tile Cholesky), panel scheme 1",
                "rc": r.returncode, "runs": lines, "stderr_tail": r.stderr[-300:] if r.returncode else ""}
    except subprocess.TimeoutExpired:
        return {"rc": "timeout", "runs": []}
    except Exception as exc:  # noqa: BLE001 - a side measurement never fails the bench
        return {"rc": f"{type(exc).__name__}: {exc}", "runs": []}


def run_sharded(args, rank, world, local_rank, n):
    """ONE exact GP of n points over all ranks (BASELINE configs[3]); returns the record on rank 0."""
    import torch

    from battgp_amd import parallel, synthetic
    from battgp_amd.sharded import make_sharded_gp

    kid, hyp, desc = kernel_setup(args.kernel)
    fault = os.environ.get("BGP_BENCH_SHARDED_FAULT", "")  # test hook (tests/test_bench_host.py): "raise:<rank>" before the first
    # collective, "late:<rank>" after the last one (a failure the other ranks can not see)
    if fault == f"raise:{rank}":
        raise RuntimeError("injected fault in the sharded sub-run")
    gp = make_sharded_gp(kid, hyp, nb=args.sharded_nb, backend_name=args.backend, local_rank=local_rank)
    x, y = synthetic.make_cell_data(n)
    xq = synthetic.make_query(x, args.m)
    steps = max(1, args.sharded_steps)
    gp.fit_predict(x, y, xq)  # warm-up: allocations, RCCL channel set-up, clocks
    gp.predict(xq)
    parallel.barrier(gp.dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        # the reference's flow: the first predict triggers the factorisation (battcellgp_full.py:171-173) - ONE fused pass,
        # the query rows ride through the panels
        lml, mean, var = gp.fit_predict(x, y, xq)
    parallel.barrier(gp.dist)
    torch.cuda.synchronize()
    red = gp.be.device if args.backend == "nccl" else "cpu"
    dt = parallel.max_over_ranks(gp.dist, time.perf_counter() - t0, device=red) / steps
    # a later prediction with other queries: the right-looking pass over the stored factor (look-ahead reduce)
    parallel.barrier(gp.dist)
    t1 = time.perf_counter()
    mean_later, _ = gp.predict(xq)
    torch.cuda.synchronize()
    later_s = parallel.max_over_ranks(gp.dist, time.perf_counter() - t1, device=red)
    later_dev = float(np.max(np.abs(mean_later - mean)) / max(np.max(np.abs(mean)), 1e-300))
    grad_s, grad = None, None
    if args.sharded_grad:  # one optimiser iteration's second half: the analytic gradient (Sigma^-1 in place over the panels)
        parallel.barrier(gp.dist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grad = gp.lml_grad()
        parallel.barrier(gp.dist)
        torch.cuda.synchronize()
        grad_s = parallel.max_over_ranks(gp.dist, time.perf_counter() - t0, device=red)
    rec = None
    if rank == 0:
        flop = algorithmic_flop(n, args.m)
        rec = {
            "workload": f"full_gp, {desc}, ONE GP of N={n} sharded over {world} GPU(s): column-panel block-cyclic Cholesky, "
                        f"nb={args.sharded_nb}, RCCL broadcast of factored panels",
            "n": n, "world": world, "steps": steps, "s_per_fit_predict": dt, "gflops": flop / dt / 1e9,
            "gflops_per_gpu": flop / dt / 1e9 / world, "frac_of_mfma_peak_per_gpu": flop / dt / 1e12 / world / PEAK_FP64_MFMA_TFLOPS,
            "timers_s": gp.timers(), "lml": lml, "jitter": gp.jitter, "mean_first": [float(v) for v in mean[:3]],
            "var_first": [float(v) for v in var[:3]], "scaling": "strong",
            "later_predict_s": later_s, "later_predict_vs_fused_max_rel": later_dev,
            # what one timed step is (records of rounds <= 3 timed fit() followed by predict(): not comparable with these)
            "flow": "fused_fit_predict",
        }
        if grad_s is not None:
            rec["lml_grad"] = {"seconds": grad_s, "tflops_per_gpu": (2.0 * n**3 / 3.0) / grad_s / 1e12 / world,
                               "over_one_fit_predict": grad_s / dt, "grad": [float(v) for v in grad]}
        # what rank 0 exchanged over all the calls above: {phase: {collective: [calls, payload B, received B]}}
        rec["comm_bytes_rank0"] = {ph: {k: list(v) for k, v in kinds.items()} for ph, kinds in gp.comm_bytes().items()}
    if fault == f"late:{rank}":
        raise RuntimeError("injected late fault in the sharded sub-run")
    gp.close()
    return rec


def self_launch(n_ranks: int) -> None:
    """A bare `python bench.py --gpus N ...` with N > 1 (no launcher above it: WORLD_SIZE unset) becomes the launch the
    contract names - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    <the command line as it was given>` - by replacing this process (exec: the launcher inherits stdout, so rank 0's ONE
    JSON line is this command's output, its exit code is this command's exit code, and a caller's SIGTERM reaches the
    launcher, which hands it to the ranks).  One process per GPU, as the reference does it (gp_runner.py:246-298).
    "The command line as it was given" = sys.orig_argv from the first script on: interpreter flags are dropped, a wrapper
    script in front of bench.py (tests/emu/run_script_emu.py in the CPU rehearsal) stays in front of every rank."""
    import socket

    orig = list(getattr(sys, "orig_argv", [])) or [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]
    start = next((i for i in range(1, len(orig)) if orig[i].endswith(".py") and os.path.isfile(orig[i])), None)
    prog = orig[start:] if start is not None else [os.path.abspath(__file__), *sys.argv[1:]]
    # torch.distributed.run's argparse takes a bare --n for an ambiguous prefix of its own options: hand it on under its other name
    prog = ["--size" if a == "--n" else "--size=" + a[4:] if a.startswith("--n=") else a for a in prog]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), *prog]
    print(f"bench.py: --gpus {n_ranks} without a launcher: exec {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--size", dest="n", type=int, default=131072,
                    help="training points per cell (configs[2]: 131 072); use --size under torch.distributed.run, whose argparse "
                         "rejects --n as an ambiguous prefix of its own options")
    ap.add_argument("--m", type=int, default=300, help="query points (battgp_full.py:98)")
    ap.add_argument("--kernel", default="matern32", choices=["battgp", "matern32"])
    ap.add_argument("--mode", default="cells", choices=["cells", "sharded", "cpu_baseline"])
    ap.add_argument("--nb", type=int, default=-1, help="outer panel width override")
    ap.add_argument("--lookahead", type=int, default=-1, help="bits 0-2: look-ahead depth (0 off, 1 default); +8: panel-stream updates ordered before rest(k); +16: no atomic epilogue")
    ap.add_argument("--panel-scheme", type=int, default=-1, help="0 = 64-wide chain over all rows, 1 = diagonal-block chain + one deep TRSM GEMM (default)")
    ap.add_argument("--slab", type=int, default=0, help="bgp_set_layout: 0 automatic, -1 full square, > 0 column-slab width")
    ap.add_argument("--cpu-n", type=int, default=16384, help="size of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-n2", type=int, default=40000, help="second CPU-baseline sample (BASELINE configs[1]'s size), run only when the first sample's rate says it fits --cpu-budget-s (0 = never)")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0)
    ap.add_argument("--no-residuals", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (steady fill, memset ceiling, N = 40 000 extra config)")
    ap.add_argument("--side-budget-s", type=float, default=540.0,
                    help="time budget of ALL side measurements after the timed region (steady fill, extra configuration, PMC passes, "
                         "schedule A/B); one that no longer fits is skipped and listed under side_measurement_errors")
    ap.add_argument("--wall-budget-s", type=float, default=900.0,
                    help="time budget of the WHOLE run: the side measurements only get what the warm-up and the timed steps have left of it")
    ap.add_argument("--extra-n", type=int, default=40000, help="size of the extra configuration measured beside the headline (BASELINE configs[1]: 40 000, the reference kernel; 0 = skip)")
    ap.add_argument("--sweep-n", type=lambda v: [int(a) for a in v.split(",") if a], default=[4096, 8192, 16384, 32768, 65536],
                    help="sizes below the headline of the `vs_n` curve (comma separated; empty = none)")
    ap.add_argument("--nmax-n", type=int, default=274432, help="size of the measured N_max fit (tools/large_n.py in a child; 0 = cite profiles/r01_large_n.json only)")
    ap.add_argument("--nmax-slab", type=int, default=2048, help="column-slab width of the N_max fit (0 = automatic)")
    ap.add_argument("--nmax-nb", type=int, default=512, help="outer panel width of the N_max fit")
    ap.add_argument("--nmax-need-s", type=float, default=150.0, help="side budget that must be left for the N_max fit to be started")
    ap.add_argument("--experiments", action="store_true", help="also run the A/B of the optional Cholesky schedules (tools/ab_lookahead.py) in a child process")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes of the headline workload (child processes, ~1-2 min)")
    ap.add_argument("--separate", action="store_true", help="bgp_fit then bgp_predict (separate triangular-solve pass) instead of the fused call")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1: nccl (= RCCL, the driver's runs); gloo with --share-gpu rehearses "
                         "the multi-rank path on a 1-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use GPU 0 (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--sharded-n", type=int, default=98304, help="size of the ONE sharded GP appended to a multi-GPU cells run (0 = skip)")
    ap.add_argument("--sharded-nb", type=int, default=1024)
    ap.add_argument("--sharded-steps", type=int, default=1)
    ap.add_argument("--sharded-grad", action="store_true", help="also time ONE analytic LML gradient of the sharded GP (2/3 N^3 flop)")
    ap.add_argument("--sharded-limit-s", type=float, default=300.0, help="time limit of the sharded sub-run of a multi-GPU cells run")
    ap.add_argument("--force-group", action="store_true",
                    help="build the process group even for ONE process, so that --mode sharded on a 1-GPU box sends every "
                         "broadcast / all-reduce of the schedule through RCCL (single-rank proxy with the collectives in)")
    args = ap.parse_args()
    import signal

    signal.signal(signal.SIGTERM, _on_term)
    signal.signal(signal.SIGINT, _on_term)
    if args.mode == "cpu_baseline":  # child of host_baseline(): host cores only, one JSON line per finished sample
        cpu_baseline(args.kernel, args.cpu_n, args.m, args.cpu_n2, args.cpu_budget_s,
                     progress=lambda rec: print(json.dumps({"cpu_baseline": rec}), flush=True))
        return
    if args.force_group:
        os.environ["BGP_FORCE_GROUP"] = "1"

    import torch

    from battgp_amd import parallel

    rank, world, local_rank = parallel.env_rank_world()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return: this process becomes the launcher
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that started {world} rank(s) (WORLD_SIZE={world}): the two must agree")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = parallel.init(args.backend, device=torch.device("cuda", local_rank))  # nccl = RCCL; None for 1 process
    red_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    hung, sh = False, None
    if args.mode == "sharded":
        rec = run_sharded(args, rank, world, local_rank, args.n)
        if rank == 0:
            flop = algorithmic_flop(args.n, args.m)
            out = {
                "metric": "exact-GP fit+predict throughput (fp64)", "value": rec["gflops"], "unit": "GFLOP/s", "n_gpus": world,
                "steps": rec["steps"], "warmup": 1, "ms_per_step": rec["s_per_fit_predict"] * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": rec["workload"], "n": args.n, "m": args.m, "kernel": args.kernel, "flop_per_step": flop,
                           "parallelism": f"1 GP sharded over {world} GPU(s)"},
                "sharded": rec,
            }
            print(json.dumps(out), flush=True)
    else:
        out = run_cells(args, rank, world, local_rank, dist, red_dev)
        sh = None
        if world > 1 and args.sharded_n > 0:
            # the cells record above is complete; whatever the sharded sub-run does on ANY rank (an exception, a collective
            # that never returns) must not cost the JSON line: it runs under a time limit and its failure is a field
            sh, hung = _guarded(lambda: run_sharded(args, rank, world, local_rank, args.sharded_n), args.sharded_limit_s, local_rank)
        if rank == 0:
            if sh is not None:
                out["sharded"] = sh
            if world == 1:
                host_baseline(out, args)
            else:
                out["cpu_baseline"] = None
            emit(out)
    if dist is not None:
        failed = hung or (isinstance(sh, dict) and "error" in sh)
        if not hung:
            # The way out is decided by ALL ranks together: a failure seen on one rank only (an exception after the
            # collectives had completed) must not let that rank leave while the others wait in the barrier.  One MAX
            # all-reduce of "something went wrong here", itself under a time limit - a rank that hung never joins it.
            def agree():
                t = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return {"failed": bool(t.item() > 0.0)}

            verdict, hung_now = _guarded(agree, 60.0, local_rank)
            failed = failed or hung_now or bool(verdict.get("failed")) or "error" in verdict
        if failed:
            # a rank of the group is stuck or gone: the tear-down collectives would wait for it - the line is out, leave
            sys.stdout.flush()
            os._exit(0)
        dist.barrier()
        dist.destroy_process_group()


def _guarded(fn, limit_s: float, local_rank: int):
    """Run ``fn`` in a worker thread for at most ``limit_s`` seconds.  Returns ``(result, hung)``; a failure or a timeout
    comes back as ``{"error": ...}`` instead of propagating (a hung HIP / RCCL call can not be interrupted: the worker is
    a daemon thread and the caller leaves with ``os._exit`` once its record is printed)."""
    import threading

    box = {}

    def body():
        try:
            import torch

            torch.cuda.set_device(local_rank)  # the current device is per thread
            box["rec"] = fn()
        except BaseException as exc:  # noqa: BLE001 - reported in the record
            box["err"] = f"{type(exc).__name__}: {exc}"[:400]

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        return {"error": f"no result within {limit_s:.0f} s (sub-run abandoned, cells record kept)"}, True
    if "err" in box:
        return {"error": box["err"]}, False
    return box["rec"], False


if __name__ == "__main__":
    main()
