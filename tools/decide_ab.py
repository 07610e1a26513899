"""Turns the A/B measurements of one GPU session into the decision DESIGN.md section 10.2 prescribes - mechanically, so that
the optional variants are either promoted or deleted the day they are measured:

    a variant becomes the default at the sizes where it is faster than the default by MORE THAN 2 % AND (schedules: bit-identical;
    fill variants: inside their elementwise bound, which the -m gpu child-process test asserts) - every other one is deleted
    together with its knob.

    python tools/decide_ab.py gpurun_out/r03s          # reads sweep_la<word>.jsonl, sweep_small_la<word>.jsonl,
                                                        # ab_lookahead.jsonl, fill_rate*.txt, bench_default.json ("experiments")
Prints one table per family and a final list "promote" / "delete".  Reads files only; needs no GPU."""
import glob
import json
import os
import re
import sys

GAIN = 1.02
LA_NAMES = {1: "default", 33: "+32 slim chain kernels", 65: "+64 split panels", 97: "+32+64 slim + split", 129: "+128 fused update+potrf",
            193: "+64+128 split + fused", 161: "+32+128 slim fused"}


def read_jsonl(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith("{"):
                try:
                    rows.append(json.loads(line))
                except json.JSONDecodeError:
                    pass
    return rows


def schedule_table(folder):
    """{la: {n: ms}} from the sweeps; identical flags from the bench's experiments"""
    ms = {}
    for path in glob.glob(os.path.join(folder, "sweep_la*.jsonl")) + glob.glob(os.path.join(folder, "sweep_small_la*.jsonl")):
        la = int(re.search(r"_la(\d+)\.jsonl$", path).group(1))
        for row in read_jsonl(path):
            ms.setdefault(la, {})[int(row["n"])] = min(float(row["fit_predict_ms"]), ms.get(la, {}).get(int(row["n"]), float("inf")))
    identical = {}
    runs = []
    ab = os.path.join(folder, "ab_lookahead.jsonl")  # tools/ab_lookahead.py (gpu_session.sh, stage "slim")
    if os.path.exists(ab):
        runs += read_jsonl(ab)
    bench = os.path.join(folder, "bench_default.json")  # bench.py --experiments files the same lines under "experiments"
    if os.path.exists(bench):
        for rec in read_jsonl(bench):
            runs += (rec.get("experiments") or {}).get("runs", [])
    for run in runs:
        la, n = int(run["lookahead"]), int(run["n"])
        identical[(la, n)] = bool(run["identical_to_default"]) and identical.get((la, n), True)
        ms.setdefault(la, {}).setdefault(n, float(run["fit_predict_ms"]))
    return ms, identical


def fill_table(folder):
    """{variant: median GB/s} from tools/fill_rate.py outputs"""
    out = {}
    for path in sorted(glob.glob(os.path.join(folder, "fill_rate*.txt"))):
        tag = os.path.basename(path)[len("fill_rate"):-len(".txt")].lstrip("_") or "default"
        rates = {}
        with open(path) as f:
            for line in f:
                m = re.match(r"(\S+)\s+N=(\d+):.*median\(2\.\.\)\s+(\d+)", line)
                if m:
                    rates[m.group(1)] = float(m.group(3))
        if rates:
            out[tag] = rates
    return out


def parity_gates(folder):
    """{"schedules": True / False / None, "fill": ...} from pytest_optional.log of the session's "optional" stage
    (tests/test_gpu_zz_optional_schedules.py, BGP_TEST_OPTIONAL=1): True = its parity cases passed, False = one failed or
    timed out, None = the log is missing / the cases did not run.  Only True lets a variant be promoted."""
    gates = {"schedules": None, "fill": None}
    path = os.path.join(folder, "pytest_optional.log")
    if not os.path.exists(path):
        return gates
    text = open(path).read()
    m = re.search(r"(\d+) passed", text)
    failed = re.findall(r"^(?:FAILED|ERROR) \S*::(\S+)", text, flags=re.M)
    ran = bool(m) or bool(failed)
    if not ran or " skipped" in text and not m and not failed:
        return gates
    sched_bad = any("test_optional_schedules_in_a_child_process" in f for f in failed)
    fill_bad = any("test_optional_interior_paths_of_the_fill" in f for f in failed)
    npass = int(m.group(1)) if m else 0
    # 2 schedule cases + 1 fill case: all three must have been seen (passed or failed) for a gate to be known
    if npass + len(failed) >= 3:
        gates["schedules"] = not sched_bad
        gates["fill"] = not fill_bad
    else:
        gates["schedules"] = False if sched_bad else None
        gates["fill"] = False if fill_bad else None
    return gates


def main(folder):
    promote, delete, undecided = [], [], []
    gates = parity_gates(folder)
    print(f"== parity gates from pytest_optional.log: {gates}  (True is needed for a promotion)")
    ms, identical = schedule_table(folder)
    base = ms.get(1, {})
    print(f"== Cholesky schedules (fit+predict ms; base = lookahead word 1), folder {folder}")
    if not base:
        print("   no sweep_la1.jsonl / experiments: nothing measured")
    for la in sorted(k for k in ms if k != 1):
        wins, cells = [], []
        for n in sorted(ms[la]):
            if n not in base:
                continue
            ratio = base[n] / ms[la][n]
            same = identical.get((la, n))
            cells.append(f"N={n}: {ms[la][n]:.2f} vs {base[n]:.2f} ({ratio:.3f}x{'' if same is None else ', identical' if same else ', DIFFERENT BITS'})")
            if ratio > GAIN and same is not False:
                wins.append(n)
            if same is False:
                wins = [w for w in wins if w != n]
        name = LA_NAMES.get(la, f"word {la}")
        print(f"   {la:4d} {name}\n        " + "\n        ".join(cells))
        bad_bits = any(identical.get((la, n)) is False for n in ms[la])
        if bad_bits:
            delete.append(f"lookahead {la} ({name}): differing bits")
        elif wins and gates["schedules"] is not True:
            undecided.append(f"lookahead {la} ({name}) is faster at N in {wins} but its parity cases "
                             f"{'FAILED' if gates['schedules'] is False else 'did not run'}: NOT promoted")
        elif wins:
            promote.append(f"lookahead {la} ({name}) at N in {wins}")
        elif cells:
            delete.append(f"lookahead {la} ({name}): never more than {100 * (GAIN - 1):.0f} % faster")
        else:
            undecided.append(f"lookahead {la} ({name}): no common size with the default")
    fills = fill_table(folder)
    print("== fill variants (steady GB/s, tools/fill_rate.py)")
    base_f = fills.get("default", {})
    for tag, rates in fills.items():
        line = ", ".join(f"{k} {v:.0f}" + (f" ({v / base_f[k]:.3f}x)" if tag != "default" and k in base_f else "") for k, v in rates.items())
        print(f"   {tag:18s} {line}")
        if tag == "default":
            continue
        gains = [v / base_f[k] for k, v in rates.items() if k in base_f]
        if not gains:
            undecided.append(f"fill variant {tag}: no default measurement beside it")
        elif max(gains) > GAIN and gates["fill"] is not True:
            undecided.append(f"fill variant {tag} is faster (best {max(gains):.3f}x) but its elementwise-bound test "
                             f"{'FAILED' if gates['fill'] is False else 'did not run'}: NOT promoted")
        elif max(gains) > GAIN:
            promote.append(f"fill variant {tag} (best {max(gains):.3f}x; elementwise-bound test green)")
        else:
            delete.append(f"fill variant {tag}: best {max(gains):.3f}x")
    print("== decision (DESIGN.md section 10.2: > 2 % faster AND inside its parity gate -> default there; otherwise deleted with its knob)")
    for title, items in (("promote", promote), ("delete", delete), ("undecided", undecided)):
        print(f"   {title}:")
        for it in items or ["(none)"]:
            print(f"     - {it}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03s"))
