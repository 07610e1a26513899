"""Summarise bench.py JSON lines from the files given (or stdin if none) - diagnostic helper."""
import fileinput, json
for line in fileinput.input():
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    p = d["phases_ms"]
    print(
        f"N={d['config']['n']} gpus={d['n_gpus']} ms/step={d['ms_per_step']:.1f} GF={d['value']:.0f} "
        f"potrf={p['potrf_ms']:.1f}ms ({d['potrf_tflops']:.1f} TF, {100*d['potrf_frac_of_peak']:.0f}%) "
        f"trail={d['roofline']['achieved']:.1f}TF fill={p['fill_ms']:.2f}ms ({d['roofline_fill']['achieved']:.0f} GB/s) "
        f"solve={p['solve_ms']:.1f} cross={p['cross_ms']:.2f} var={p['var_ms']:.1f} resid={d.get('residuals')}"
    )
