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
