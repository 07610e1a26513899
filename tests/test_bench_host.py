"""Host-side logic of bench.py that needs no GPU: the reduction of the live rocprofv3 --pmc passes (pmc_live) against a
stand-in profiler that writes counter_collection CSVs with known values."""
import json
import os
import stat
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE = textwrap.dedent('''\
    #!{py}
    """stand-in for rocprofv3: --pmc C... --output-format csv -d DIR -o NAME -- cmd...; writes DIR/host/NAME_counter_collection.csv"""
    import csv, os, sys
    a = sys.argv[1:]
    counters = a[a.index("--pmc") + 1:a.index("--output-format")]
    d, name = a[a.index("-d") + 1], a[a.index("-o") + 1]
    assert "--kernel-trace" not in a and "--sys-trace" not in a, "no trace domain next to --pmc"
    n = int(a[a.index("--") + 3])
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    rows = [("void (anonymous namespace)::gemm_nt_kernel<128, 128, 2, 0>(double*, long)", 3),
            ("(anonymous namespace)::gemv_t_partial_kernel(double const*, long)", 1),
            ("void (anonymous namespace)::fill_kernel<2, 4, 0>(FillParams)", 2)]
    val = {{"FETCH_SIZE": {{0: 1000.0, 1: float(os.environ["FAKE_GEMV_KIB"]), 2: 5.0}},
           "WRITE_SIZE": {{0: 400.0, 1: 1.0, 2: float(os.environ["FAKE_FILL_KIB"]) / 2}},
           "SQ_VALU_MFMA_BUSY_CYCLES": {{0: 0.75 * 1024 * 1000, 1: 0.0, 2: 0.0}},
           "GRBM_GUI_ACTIVE": {{0: 8 * 1000.0, 1: 8 * 10.0, 2: 8 * 10.0}}}}
    with open(os.path.join(d, "host", name + "_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        i = 0
        for k, (kn, reps) in enumerate(rows):
            for _ in range(reps):
                for c in counters:
                    w.writerow([i, kn, c, val[c][k], 1000 * i, 1000 * i + 500])
                i += 1
    ''')


def test_pmc_live_reduction_with_a_stand_in_profiler(tmp_path, monkeypatch):
    import bench

    n, m = 2048, 300
    fake = tmp_path / "rocprofv3"
    fake.write_text(FAKE.format(py=sys.executable))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    nb = 512
    gemv_expected = sum(8.0 * (n - min(k0 + nb, n)) * nb for k0 in range(0, n, nb))
    fill_alg = 4.0 * n * (n + 1) + 8.0 * m * n
    monkeypatch.setenv("BGP_ROCPROFV3", str(fake))
    monkeypatch.setenv("FAKE_GEMV_KIB", repr(gemv_expected / 1024 / 2))  # the counter tallies half of a wide read
    monkeypatch.setenv("FAKE_FILL_KIB", repr(fill_alg / 1024))
    out = bench.pmc_live(n, "matern32", m, limit_s=60)
    assert all(p["rc"] == 0 for p in out["passes"].values()), out
    assert out["dispatches"] == 3
    assert out["fetch_bytes_corrected_per_dispatch"] == pytest.approx(1000.0 * 1024 * 2)
    assert out["write_bytes_per_dispatch"] == pytest.approx(400.0 * 1024)
    assert out["hbm_bytes_per_dispatch"] == pytest.approx(2400.0 * 1024)
    assert out["hbm_bytes_total"] == pytest.approx(3 * 2400.0 * 1024)
    cal = out["calibration"]
    assert [v for k, v in cal.items() if k.startswith("fetch_raw")][0] == pytest.approx(0.5)
    assert cal["fill_write_over_algorithmic"] == pytest.approx(1.0)
    assert out["mfma_util"] == pytest.approx(0.75)
    assert out["effective_clock_ghz_under_pmc"] == pytest.approx(1000.0 / 500.0)
    json.dumps(out)  # goes into the bench line


def test_pmc_live_reports_a_missing_or_failing_profiler(tmp_path, monkeypatch):
    import bench

    monkeypatch.setenv("BGP_ROCPROFV3", str(tmp_path / "absent"))
    assert bench.pmc_live(2048, "battgp")["rc"] == "rocprofv3 not found"
    bad = tmp_path / "rocprofv3"
    bad.write_text("#!/bin/sh\nexit 7\n")
    bad.chmod(0o755)
    monkeypatch.setenv("BGP_ROCPROFV3", str(bad))
    out = bench.pmc_live(2048, "battgp", limit_s=30)
    assert all(p["rc"] == 7 for p in out["passes"].values())
    assert "hbm_bytes_per_dispatch" not in out and "mfma_util" not in out


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_the_drivers_multi_gpu_launch_of_bench_py_rehearsed_on_the_cpu_build():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --steps K --warmup W` - the launch the
    driver uses for SCALE, which no GPU has executed yet - with the CPU build of the kernel sources behind the binding
    (tests/emu/run_script_emu.py) and gloo between the ranks: rendezvous, barriers, max-over-ranks timing, one cell per
    rank, the appended `sharded` sub-record (ONE GP over both ranks) and the single JSON line of rank 0.  The numbers it
    prints are not measurements; the LML values in it are checked against the oracle."""
    import subprocess

    from battgp_amd import KERNEL_MATERN32, synthetic
    from oracle.exact_gp import OracleGP

    n, ns = 600, 384
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py",
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", str(n), "--backend", "gloo", "--share-gpu",
           "--sharded-n", str(ns), "--sharded-nb", "128"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "GFLOP/s" and out["dtype"] == "f64" and out["vs_baseline"] is None and out["cpu_baseline"] is None
    assert out["config"]["parallelism"] == "2 independent cells" and "workload" in out["config"]
    # whole-job value: both cells' flop over the slowest rank's time
    assert out["value"] == pytest.approx(2 * out["config"]["flop_per_step"] / (out["ms_per_step"] * 1e-3) / 1e9, rel=1e-9)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    x, y = synthetic.make_cell_data(n, seed=n)  # rank 0's cell
    assert out["lml"] == pytest.approx(OracleGP(KERNEL_MATERN32, synthetic.HYP_MATERN32, x, y).fit().lml, rel=1e-9)
    sh = out["sharded"]
    assert sh["world"] == 2 and sh["n"] == ns and sh["scaling"] == "strong"
    xs, ys = synthetic.make_cell_data(ns)
    assert sh["lml"] == pytest.approx(OracleGP(KERNEL_MATERN32, synthetic.HYP_MATERN32, xs, ys).fit().lml, rel=1e-9)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_a_bare_multi_gpu_invocation_launches_itself_under_torch_distributed_run():
    """`python3 bench.py --gpus 2 ...` with NO launcher above it (WORLD_SIZE unset - the shape of the
    driver's 1-GPU command with another N) must not exit: it re-executes itself under `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1` and rank 0 prints the ONE JSON line.  One process per GPU is the
    reference's own shape (gp_runner.py:246-298).  Rehearsed on the CPU build (the wrapper stays in front of every rank)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py", "--gpus", "2", "--backend", "gloo", "--share-gpu",
           "--steps", "1", "--warmup", "0", "--n", "300", "--sharded-n", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["config"]["n"] == 300 and out["value"] > 0
    assert out["config"]["parallelism"] == "2 independent cells"


def test_self_launch_command_line(monkeypatch, tmp_path):
    """bench.self_launch: the launcher line is the contract's, the script's own arguments follow unchanged (--n under its
    other name: torch.distributed.run's parser trips over it), interpreter flags are dropped"""
    import bench

    seen = {}
    monkeypatch.setattr(os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv))
    monkeypatch.setattr(sys, "orig_argv", [sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--n", "4096", "--steps", "2"])
    bench.self_launch(4)
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[:6] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4"]
    assert a[6:8] == ["--master-addr", "127.0.0.1"] and a[8] == "--master-port" and 1024 < int(a[9]) < 65536
    assert a[10:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "4096", "--steps", "2"]


def test_guarded_sub_run_reports_instead_of_propagating(monkeypatch):
    """bench._guarded: result, exception and time-out of the sharded sub-run all come back as a value"""
    import time

    import torch

    import bench

    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    assert bench._guarded(lambda: {"ok": 1}, 5.0, 0) == ({"ok": 1}, False)
    rec, hung = bench._guarded(lambda: 1 / 0, 5.0, 0)
    assert "ZeroDivisionError" in rec["error"] and not hung
    rec, hung = bench._guarded(lambda: time.sleep(30), 0.2, 0)
    assert hung and "no result within" in rec["error"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
@pytest.mark.parametrize("fault", ["raise:1", "late:0", "late:1"])
def test_a_failing_sharded_sub_run_cannot_take_the_cells_record_down(fault):
    """the multi-GPU launch with rank 1 raising inside the sharded sub-run (rank 0 then waits in a collective
    that never completes): the cells record - already measured - is still printed as the ONE JSON line, with the failure
    as a field, and both ranks leave without the tear-down collectives.  ADVICE r3: a failure only ONE rank sees, after
    the last collective of the sub-run ("late"), must not let that rank leave while the other waits in the barrier until
    the communicator's time-out: the ranks agree on the way out (one MAX all-reduce under a time limit)."""
    import subprocess
    import time

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py",
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "300", "--backend", "gloo", "--share-gpu",
           "--sharded-n", "256", "--sharded-nb", "128", "--sharded-limit-s", "15"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, BGP_BENCH_SHARDED_FAULT=fault), capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    if fault == "late:1":  # rank 0's record of the sub-run was complete; the ranks still leave together
        assert out["sharded"]["lml"] and "comm_bytes_rank0" in out["sharded"]
    else:
        assert "error" in out["sharded"], out["sharded"]
    if fault.startswith("late"):
        assert r.returncode == 0 and time.perf_counter() - t0 < 240, (r.returncode, time.perf_counter() - t0)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_side_measurements_cannot_take_the_bench_record_down(tmp_path):
    """the default single-GPU invocation rehearsed on the CPU build, where several side measurements MUST fail (no
    device for torch's memset timing, no profiler, no GPU for the A/B child): the record of the timed steps is printed
    all the same, with what did run (steady fill, the extra configuration) and the failures listed"""
    import subprocess

    env = dict(os.environ, BGP_ROCPROFV3=str(tmp_path / "absent"))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py", "--steps", "1", "--warmup", "0", "--size", "600",
           "--extra-n", "500", "--cpu-n", "300", "--experiments", "--sweep-n", "256,400,600,900", "--nmax-n", "1500", "--nmax-need-s", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["n"] == 600 and out["lml"] == out["lml"]
    assert out["roofline_fill"]["steady_gbs"] > 0                      # ran
    assert "fill_steady" in out["side_measurement_errors"]             # torch's event timing has no device here
    assert out["pmc_live"] == {"rc": "rocprofv3 not found"} and out["roofline"]["traffic"] is None
    assert out["experiments"]["rc"] != 0 and out["experiments"]["runs"] == []
    assert [e["workload"].split(", N=")[1].split(" ")[0] for e in out["extra_configs"]] == ["500"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and set(cb["phases"]) == {"fill_s", "potrf_s", "solve_predict_s"}
    # BASELINE's "vs N" curve: the sizes below the headline, then the headline point from the timed region
    assert [p["n"] for p in out["vs_n"]] == [256, 400, 600] and out["vs_n"][-1]["note"].startswith("the timed region")
    assert all({"n", "ms", "gflops", "frac_of_peak", "lml"} <= set(p) and p["ms"] > 0 for p in out["vs_n"])
    assert out["vs_n"][-1]["gflops"] == pytest.approx(out["value"]) and out["vs_n"][-1]["lml"] == out["lml"]
    # the N_max child has no GPU here: the citation stays, labelled as one, with the reason
    nm = out["n_max_per_gpu"]
    assert nm["source"].startswith("CITED") and "large_n.py" in nm["fallback_because"] and "n_max" in out["side_measurement_errors"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_smoke_entry_point_rehearsed_on_the_cpu_build():
    """__graft_entry__.smoke() - what the driver runs first on the GPU box - through the CPU build of the kernel
    sources: engine call, oracle comparison and the plugin surface on top, end to end"""
    import subprocess

    code = ("import sys; sys.path.insert(0, 'tests/emu'); from inject import fake_cuda_tensors, installed; fake_cuda_tensors()\n"
            "import __graft_entry__ as g\n"
            "with installed():\n    g.smoke()\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
@pytest.mark.parametrize("argv,expect", [
    (["tools/sweep_n.py", "1", "300"], '"fit_predict_ms"'),
    (["tools/system_probe.py", "300"], '"predict_all_ms"'),
    (["tools/train_iter.py", "300"], '"refit_plus_grad_ms"'),
    (["tools/fill_rate.py", "1024"], "median"),
    (["tools/profile_workload.py", "700", "matern32"], '"alpha_norm"'),
    (["tools/ab_lookahead.py", "1", "700"], '"identical_to_default": true'),
    (["tools/large_n.py", "1500"], '"fit_predict_s"'),
    (["bench.py", "--mode", "sharded", "--size", "700", "--kernel", "battgp", "--steps", "1", "--warmup", "1", "--cpu-n", "0",
      "--no-extras", "--backend", "gloo", "--force-group", "--sharded-nb", "128", "--sharded-grad"], '"lml_grad"'),
])
def test_gpu_session_scripts_rehearsed_on_the_cpu_build(argv, expect):
    """every script tools/gpu_session.sh spends GPU minutes on runs to completion on the CPU build of the kernel sources
    (control flow and output format only - a Python error must not be what the first GPU minutes find)"""
    import subprocess

    env = dict(os.environ, BGP_ONLY="battgp")
    if argv[0] == "tools/ab_lookahead.py":  # the A/B of the optional schedules: the experimental configuration of the CPU build
        env["BGP_EMU_EXPERIMENTAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), *argv], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and expect in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    if argv[0] == "tools/ab_lookahead.py":
        assert '"identical_to_default": false' not in r.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_side_measurements_respect_their_time_budget():
    """--side-budget-s: a side measurement that no longer fits is not started and is listed as skipped"""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py", "--steps", "1", "--warmup", "0",
           "--size", "700", "--extra-n", "300", "--cpu-n", "0", "--side-budget-s", "0", "--experiments"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    errs = out["side_measurement_errors"]
    assert set(errs) == {"fill_steady", "extra_300_battgp", "pmc_live", "experiments", "n_max", *(f"vs_n_{v}" for v in (4096, 8192, 16384, 32768, 65536) if v < 700)}
    assert all(v.startswith("skipped") for v in errs.values())
    assert out["extra_configs"] == [] and out["pmc_live"] is None and out["experiments"] is None
    assert [p["n"] for p in out["vs_n"]] == [700] and out["n_max_per_gpu"]["fallback_because"].startswith("skipped")
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_a_time_limit_during_the_side_measurements_still_prints_the_record(tmp_path):
    """the caller's time limit (SIGTERM) arriving while a profiler pass is running: the ONE JSON line comes out with the
    timed region, the roofline objects and the host baseline that were finished by then, flagged `interrupted`, and the
    profiler's process tree (its own session) is gone"""
    import signal
    import subprocess
    import time

    marker = tmp_path / "started"
    slow = tmp_path / "rocprofv3"
    slow.write_text(f"#!/bin/sh\necho $$ > {marker}\nsleep 300 &\necho $! >> {marker}\nwait\n")
    slow.chmod(0o755)
    env = dict(os.environ, BGP_ROCPROFV3=str(slow))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "bench.py", "--steps", "1", "--warmup", "0", "--size", "600",
           "--extra-n", "0", "--cpu-n", "300", "--cpu-n2", "0"]
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        t0 = time.time()
        while time.time() - t0 < 500 and not (marker.exists() and len(marker.read_text().split()) == 2):
            assert p.poll() is None, p.communicate()
            time.sleep(0.2)
        pids = [int(v) for v in marker.read_text().split()]
        p.send_signal(signal.SIGTERM)
        so, se = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 0, so[-1500:] + se[-1500:]
    lines = [ln for ln in so.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert "signal 15" in out["interrupted"] and out["value"] > 0 and out["roofline"]["frac"] > 0 and out["roofline_fill"]["achieved"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0  # gathered BEFORE the profiler passes
    time.sleep(0.5)
    for pid in pids:  # the stand-in profiler and its child were killed with their session
        alive = True
        try:
            os.kill(pid, 0)
            with open(f"/proc/{pid}/stat") as f:
                alive = f.read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs the ROCm host clang to build the CPU stand-in")
def test_n_max_size_follows_the_free_hbm():
    """ADVICE r05 (bench.py N_max child): the default request, N = 274 432 in slabs of 2048, needs 306.76 GB of the 308.56 GB an
    EMPTY MI355X has (profiles/r01_large_n.json) - next to the parent's own HIP context it would not fit.  The size is stepped
    down to what the free HBM holds with 3 GB to spare; the memory model is held against the two measured records."""
    import bench

    runs = {(r["n"], r["slab_width"], r["nb_outer"]): r["device_bytes"] for r in json.load(open(os.path.join(ROOT, "profiles", "r01_large_n.json")))["runs"]}
    assert bench.n_max_need_bytes(274432, 300, 2048, 512) == pytest.approx(runs[(274432, 2048, 512)], rel=2e-3)
    got, want = bench.n_max_need_bytes(262144, 300, 16384, 512), runs[(262144, 16384, 512)]
    assert want <= got <= 1.01 * want  # wide slabs: over-estimated by < 1 %, the safe side
    empty = 308556070912  # hbm_free_before of those records
    assert bench.n_max_size_for(empty, 274432, 300, 2048, 512, margin_b=0) == 274432  # what round 1 ran, on an empty GPU
    n = bench.n_max_size_for(empty - 1.0e9, 274432, 300, 2048, 512)  # next to a parent process holding ~1 GB
    assert 262144 <= n < 274432 and n % 4096 == 0
    assert bench.n_max_need_bytes(n, 300, 2048, 512) + 3e9 <= empty - 1.0e9 < bench.n_max_need_bytes(n + 4096, 300, 2048, 512) + 3e9
    assert bench.n_max_size_for(284e9, 274432, 300, 2048, 512) == 262144   # 280.2 GB in slabs of 2048 + the margin
    assert bench.n_max_size_for(100e9, 274432, 300, 2048, 512) == 0        # below the full-square ceiling: cite, do not measure
    assert bench.n_max_size_for(48 << 30, 1500, 300, 512, 128) == 1500     # a rehearsal-sized request is passed through
    # the "vs N" points ask the side budget for what they cost: N = 65 536 is ~10 s, not 5
    assert bench.vs_n_need_s(4096, 300) < 3.1 and 10.0 < bench.vs_n_need_s(65536, 300) < 15.0


def test_measured_n_max_record_is_built_from_the_large_n_child(monkeypatch):
    """bench.n_max_measured: the child is tools/large_n.py with the slab layout given; its JSON line (produced here by the
    same script on the CPU build, at a toy size) becomes the `n_max_per_gpu` object, labelled as measured; a failing child
    raises (-> side_measurement_errors, the citation stays)"""
    import subprocess

    import bench

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_script_emu.py"), "tools/large_n.py", "1500", "matern32", "512", "300", "128", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    seen = {}

    def fake_child(cmd, timeout, **kw):
        seen["cmd"], seen["timeout"] = cmd, timeout
        return subprocess.CompletedProcess(cmd, 0, r.stdout, "")

    monkeypatch.setattr(bench, "run_child", fake_child)
    rec = bench.n_max_measured(1500, "matern32", 300, 512, 128, 77.0)
    assert seen["cmd"][1:] == [os.path.join(ROOT, "tools", "large_n.py"), "1500", "matern32", "512", "300", "128", "1"] and seen["timeout"] == 77.0
    assert rec["n"] == 1500 and rec["slab_width"] == 512 and rec["nb_outer"] == 128 and rec["source"].startswith("measured by this run")
    assert rec["frac_of_peak"] == pytest.approx(rec["gflops"] / 1e3 / 78.6) and rec["residuals"]["rel_solve"] < 1e-8
    json.dumps(rec)
    monkeypatch.setattr(bench, "run_child", lambda cmd, timeout, **kw: subprocess.CompletedProcess(cmd, 1, "", "hipErrorOutOfMemory"))
    with pytest.raises(RuntimeError, match="hipErrorOutOfMemory"):
        bench.n_max_measured(1500, "matern32", 300, 512, 128, 77.0)


def test_ab_decision_tool_applies_the_rule(tmp_path):
    """tools/decide_ab.py: > 2 % faster AND bit-identical -> promote at those sizes; differing bits or no gain -> delete"""
    import subprocess

    def w(name, rows):
        (tmp_path / name).write_text("\n".join(json.dumps(r) for r in rows) + "\n")

    w("sweep_la1.jsonl", [{"n": 4096, "fit_predict_ms": 3.7}, {"n": 16384, "fit_predict_ms": 32.0}])
    w("sweep_la33.jsonl", [{"n": 4096, "fit_predict_ms": 3.9}, {"n": 16384, "fit_predict_ms": 29.0}])
    w("sweep_la129.jsonl", [{"n": 4096, "fit_predict_ms": 3.69}, {"n": 16384, "fit_predict_ms": 31.9}])
    w("bench_default.json", [{"experiments": {"runs": [{"n": 16384, "lookahead": 33, "fit_predict_ms": 29.1, "identical_to_default": True},
                                                       {"n": 16384, "lookahead": 65, "fit_predict_ms": 25.0, "identical_to_default": False},
                                                       {"n": 16384, "lookahead": 1, "fit_predict_ms": 32.0, "identical_to_default": True}]}}])
    (tmp_path / "fill_rate.txt").write_text("matern32  N=131072: 4000 5500 5510  GB/s   median(2..) 5507\n")
    (tmp_path / "fill_rate_table256.txt").write_text("matern32  N=131072: 4000 5900 5910  GB/s   median(2..) 5905\n")
    (tmp_path / "fill_rate_mfma.txt").write_text("matern32  N=131072: 4000 5500 5510  GB/s   median(2..) 5520\n")

    def decide():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "decide_ab.py"), str(tmp_path)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        promote, rest = r.stdout.split("promote:")[1].split("delete:")
        delete, undecided = rest.split("undecided:")
        return promote, delete, undecided

    # no promotion without the parity cases of the optional stage having PASSED (pytest_optional.log)
    promote, delete, undecided = decide()
    assert "lookahead 33" not in promote and "table256" not in promote
    assert "lookahead 33" in undecided and "did not run" in undecided and "table256" in undecided
    (tmp_path / "pytest_optional.log").write_text("..F\nFAILED tests/test_gpu_zz_optional_schedules.py::test_optional_interior_paths_of_the_fill_on_the_gpu - x\n1 failed, 2 passed in 9s\n")
    promote, delete, undecided = decide()
    assert "lookahead 33" in promote and "table256" not in promote and "FAILED" in undecided
    (tmp_path / "pytest_optional.log").write_text("...\n3 passed in 9s\n")
    promote, delete, undecided = decide()
    assert "lookahead 33" in promote and "[16384]" in promote and "table256" in promote
    assert "lookahead 65" in delete and "differing bits" in delete and "lookahead 129" in delete and "fill variant mfma" in delete
