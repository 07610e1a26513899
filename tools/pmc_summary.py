"""Summarise the rocprofv3 passes of tools/profile_round.sh (kernel stats + FETCH_SIZE / WRITE_SIZE /
MFMA counters, each collected in a pass of its own) into profiles/<tag>_n<N>_summary.json and copy the
kernel stats next to it.      python tools/pmc_summary.py gpurun_out/r1g r01 40000

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a coalesced streaming read.  Calibrated in the same run: gemv_t_partial reads
the strictly-lower panels of L exactly once; the fill writes 4 N (N+1) + 8 M N bytes exactly once.
MFMA utilisation = rocprofv3's MfmaUtil expression, sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per
XCC * 1024 SIMDs); a v_mfma_f64_16x16x4_f64 holds its SIMD's matrix pipe for 64 cycles."""
import csv
import json
import os
import shutil
import sys

src, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
kname = sys.argv[4] if len(sys.argv) > 4 else "battgp"
suffix = "" if kname == "battgp" else "_" + kname
m, simds, xccs = 300, 1024, 8
nb = 1024 if (n + 63) // 64 * 64 >= 32768 else 512  # the engine's default outer panel width (apply_auto_nb)


def load(p):
    with open(os.path.join(src, p, "per_kernel.json")) as f:
        return json.load(f)


def pick(d, frag, counter):
    ks = [k for k in d if frag in k]
    return sum(d[k][counter]["dispatches"] for k in ks), sum(d[k][counter]["sum"] for k in ks)


fetch, write, mfma, mops = load("pmc_fetch"), load("pmc_write"), load("pmc_mfma"), load("pmc_mops")
stats = list(csv.DictReader(open(os.path.join(src, "stats", "st_kernel_stats.csv"))))
KIB = 1024.0
npad = (n + 63) // 64 * 64
# gemv_t_partial: panel b reads rows below it, (npad - K1_b) x nbk doubles
gemv_expected = sum(8.0 * (npad - min(k0 + nb, npad)) * (min(k0 + nb, npad) - k0) for k0 in range(0, npad, nb))
_, gemv_fetch = pick(fetch, "gemv_t_partial_kernel", "FETCH_SIZE")
fetch_factor = 2.0
fill_calls, fill_write = pick(write, "fill_kernel", "WRITE_SIZE")
fill_alg = 4.0 * n * (n + 1) + 8.0 * m * n

classes = {}
for name, frag in (("trailing_update_gemm_nt_128x128_mode2", "gemm_nt_kernel<128, 128, 2"),
                   ("panel_update_gemm_nt_128x128_mode0", "gemm_nt_kernel<128, 128, 0"),
                   ("chain_update_gemm_nt_64x64_mode0", "gemm_nt_kernel<64, 64, 0"),
                   ("chain_trsm_gemm_nt_64x64_mode1", "gemm_nt_kernel<64, 64, 1"),
                   ("trsm_by_panel_inverse_gemm_nt_128x64_mode1", "gemm_nt_kernel<128, 64, 1")):
    calls, f = pick(fetch, frag, "FETCH_SIZE")
    _, w = pick(write, frag, "WRITE_SIZE")
    _, busy = pick(mfma, frag, "SQ_VALU_MFMA_BUSY_CYCLES")
    _, gui = pick(mfma, frag, "GRBM_GUI_ACTIVE")
    _, mo = pick(mops, frag, "SQ_INSTS_VALU_MFMA_MOPS_F64")
    _, wc = pick(mops, frag, "SQ_WAVE_CYCLES")
    _, wi = pick(mops, frag, "SQ_WAIT_INST_ANY")
    dur = sum(float(r["TotalDurationNs"]) for r in stats if frag in r["Name"])
    classes[name] = {
        "dispatches": calls,
        "total_duration_ms_unprofiled_pass": dur * 1e-6,
        "fetch_bytes_corrected": f * KIB * fetch_factor,
        "write_bytes": w * KIB,
        "hbm_bytes_total": f * KIB * fetch_factor + w * KIB,
        "hbm_bytes_per_dispatch": (f * KIB * fetch_factor + w * KIB) / max(1, calls),
        "mfma_flop_executed": mo * 512.0,
        "mfma_busy_cycles": busy,
        "mfma_util": busy / (gui / xccs * simds) if gui else None,
        "effective_clock_ghz": (gui / xccs) / dur if dur else None,
        "wait_inst_any_over_wave_cycles": wi / wc if wc else None,
    }
tr = classes["trailing_update_gemm_nt_128x128_mode2"]
# algorithmic C traffic of the rank-NB updates: every trailing element read and written once per panel
tr["algorithmic_c_traffic_bytes"] = 2 * 4.0 * n**3 / (3 * nb)
tr["algorithmic_flop"] = sum((npad - k1) * (npad - k1 + 1.0) * nb for k1 in range(nb, npad, nb))

out = {
    "source": src,
    "workload": f"tools/profile_workload.py {n}: one fused fit+predict ({kname}, M = {m}) + alpha()",
    "calibration": {
        "fetch_raw_over_expected_gemv_t": gemv_fetch * KIB / gemv_expected,
        "fetch_correction_factor": fetch_factor,
        "fill_write_over_algorithmic": fill_write * KIB / fill_alg,
    },
    "gemm_nt_128x128": tr,  # key read by bench.py
    "classes": classes,
    "fill": {"dispatches": fill_calls, "write_bytes": fill_write * KIB, "algorithmic_bytes": fill_alg},
    "kernel_stats": [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in stats[:14]],
}
os.makedirs("profiles", exist_ok=True)
with open(os.path.join("profiles", f"{tag}_n{n}{suffix}_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
shutil.copy(os.path.join(src, "stats", "st_kernel_stats.csv"), os.path.join("profiles", f"{tag}_n{n}{suffix}_kernel_stats.csv"))
for p in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_mops"):
    shutil.copy(os.path.join(src, p, "per_kernel.json"), os.path.join("profiles", f"{tag}_n{n}{suffix}_{p}_per_kernel.json"))
print(json.dumps(out["calibration"]))
print(json.dumps({k: (v["mfma_util"], v["hbm_bytes_per_dispatch"], v["effective_clock_ghz"]) for k, v in classes.items()}))
