"""Summarise rocprofv3 outputs (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into
profiles/<tag>_summary.json.   python tools/pmc_summary.py gpurun_out/r1 r01 40000

FETCH_SIZE correction (MI355X_MICROARCH.md section HBM): on gfx950 it reports 1/2 of the bytes of a
coalesced streaming read - calibrated here on gemv_n_sub/gemv_t_partial, which read L exactly once
(4 N^2 bytes each): factor 2.  WRITE_SIZE is calibrated on the fill kernel (4 N (N+1) bytes): factor 1."""
import collections
import csv
import json
import os
import sys

src, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        a[k][0] += 1
        a[k][1] += float(r["Counter_Value"]) * 1024.0
    return a


fetch = agg(os.path.join(src, "pmc_fetch", "pf_counter_collection.csv"))
write = agg(os.path.join(src, "pmc_write", "pw_counter_collection.csv"))
stats = list(csv.DictReader(open(os.path.join(src, "stats", "st_kernel_stats.csv"))))


def pick(d, frag):
    ks = [k for k in d if frag in k]
    return sum(d[k][0] for k in ks), sum(d[k][1] for k in ks)


cal_calls, cal_fetch = pick(fetch, "gemv_n_sub_kernel")
cal2_calls, cal2_fetch = pick(fetch, "gemv_t_partial_kernel")
l_bytes = 4.0 * n * n  # lower triangle read once
fetch_factor = 2.0
fill_calls, fill_write = pick(write, "fill_kernel")
gemm_calls, gemm_fetch = pick(fetch, "gemm_nt_kernel<128, 128, ")
_, gemm_write = pick(write, "gemm_nt_kernel<128, 128, ")
out = {
    "source": src,
    "workload": f"bench.py --n {n} --steps 1 --warmup 0 (one fit + predict, K0)",
    "calibration": {
        "fetch_raw_over_expected_gemv_n": cal_fetch / l_bytes,
        "fetch_raw_over_expected_gemv_t": cal2_fetch / l_bytes,
        "fetch_correction_factor": fetch_factor,
        "fill_write_over_algorithmic": fill_write / (4.0 * n * (n + 1)),
    },
    "gemm_nt_128x128": {
        "dispatches": gemm_calls,
        "fetch_bytes_corrected": gemm_fetch * fetch_factor,
        "write_bytes": gemm_write,
        "hbm_bytes_total": gemm_fetch * fetch_factor + gemm_write,
        "hbm_bytes_per_dispatch": (gemm_fetch * fetch_factor + gemm_write) / max(1, gemm_calls),
        "algorithmic_c_traffic_bytes": 2 * 4.0 * n**3 / (3 * 512),
    },
    "fill": {"dispatches": fill_calls, "write_bytes": fill_write, "algorithmic_bytes": 4.0 * n * (n + 1)},
    "kernel_stats": [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")} for r in stats[:12]],
}
os.makedirs("profiles", exist_ok=True)
with open(os.path.join("profiles", f"{tag}_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["calibration"]), json.dumps(out["gemm_nt_128x128"]))
