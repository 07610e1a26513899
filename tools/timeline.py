"""Timeline of one factorisation from a rocprofv3 --kernel-trace CSV: where the main stream waits.

    python tools/timeline.py <..._kernel_trace.csv> [nb=512]

Prints, per outer panel k: duration of rest(k) and la(k) (gemm MODE 2), the span of panel(k+1)'s 24-launch
chain on the panel stream, and the idle gap on the main stream before rest(k+1)."""
import csv
import json
import sys

path = sys.argv[1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", ""),
                     int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
# several fits in one trace: keep the last one (from its training fill on)
fills = [r for r in rows if "fill_kernel" in r[2]]
if fills:
    big = max(r[5] for r in fills)
    start = max(r[0] for r in fills if r[5] == big)
    rows = [r for r in rows if r[0] >= start]


def kind(name):
    if "gemm_nt_kernel<128, 128, 2" in name:
        return "deep"
    if "gemm_nt_kernel<128, 128, 0" in name or "gemm_nt_kernel<64, 64, 0" in name or "chain_gemm_slim_kernel<0" in name:
        return "upd"
    if "gemm_nt_kernel<128, 64, 1" in name or "gemm_nt_kernel<64, 64, 1" in name or "chain_gemm_slim_kernel<1" in name:
        return "trsm"
    if "potrf_tile" in name:
        return "tile"
    if "fill_kernel" in name:
        return "fill"
    return "other"


deep = [r for r in rows if kind(r[2]) == "deep"]
streams = {}
for r in deep:
    streams.setdefault(r[4] or r[3], []).append(r)
# the main stream carries the bigger launches
main_id = max(streams, key=lambda s: sum(r[5] for r in streams[s]))
rest = streams[main_id]
la = [r for s, v in streams.items() if s != main_id for r in v]
panel = [r for r in rows if kind(r[2]) in ("tile", "trsm", "upd")]
busy = sum(r[1] - r[0] for r in rest)
span = rest[-1][1] - rest[0][0]
gaps = [(rest[i + 1][0] - rest[i][1]) for i in range(len(rest) - 1)]
first_tile = min(r[0] for r in panel)
last_end = max(r[1] for r in rows if kind(r[2]) != "other")
out = {
    "n_rest": len(rest), "n_la": len(la),
    "total_ms": (last_end - first_tile) / 1e6,
    "head_ms (first tile -> first rest)": (rest[0][0] - first_tile) / 1e6,
    "rest_busy_ms": busy / 1e6, "rest_span_ms": span / 1e6, "gap_sum_ms": sum(g for g in gaps if g > 0) / 1e6,
    "tail_ms (last rest end -> end)": (last_end - rest[-1][1]) / 1e6,
    "la_busy_ms": sum(r[1] - r[0] for r in la) / 1e6,
    "panel_kernels_busy_ms": sum(r[1] - r[0] for r in panel) / 1e6,
}
print(json.dumps(out, indent=1))
print("  k   rest_ms   gap_after_us   chain_in_window: n_kernels  span_us  busy_us")
for i, r in enumerate(rest):
    nxt = rest[i + 1][0] if i + 1 < len(rest) else last_end
    win = [p for p in panel if p[0] >= r[0] and p[0] < nxt]
    sp = (max(p[1] for p in win) - min(p[0] for p in win)) / 1e3 if win else 0
    bz = sum(p[1] - p[0] for p in win) / 1e3
    if i < 6 or i % 8 == 0 or i > len(rest) - 12:
        print(f"{i:3d} {(r[1]-r[0])/1e6:9.3f} {(nxt - r[1])/1e3:12.1f} {len(win):10d} {sp:9.1f} {bz:9.1f}")
