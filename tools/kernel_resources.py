"""Static facts about the shipped kernels, from the gfx950 assembly hipcc writes (no GPU needed):

  * the resource table (VGPR / AGPR / LDS / scratch per kernel, from the code-object metadata): what decides how many
    workgroups of which kernels can share a CU;
  * the instruction mix of a kernel's hot loop (`loop_profile`): MFMA / LDS reads / global loads / barriers inside the loop
    that holds the most MFMAs - the k-loop of the GEMM kernels.

    python tools/kernel_resources.py            -> the table
    python tools/kernel_resources.py --loops    -> + the k-loop mix of every gemm_nt_kernel instantiation
    python tools/kernel_resources.py --experimental   -> of the library built with -DBGP_EXPERIMENTAL

tests/test_kernel_static.py turns both into a gate of the CPU suite (the class of fault the CPU build of the
kernel sources cannot see - spills, a register count that halves the occupancy, an instantiation whose main loop differs)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "battgp_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return [re.sub(r"^void ", "", re.sub(r"\(.*", "", o.replace("(anonymous namespace)::", ""))) for o in out]
        except (FileNotFoundError, subprocess.CalledProcessError):
            continue
    return names


def compile_asm(src_name: str, out_dir: str, defines=()) -> str:
    """gfx950 device assembly of battgp_amd/csrc/<src_name>, same flags as battgp_amd/build.py; returns its text"""
    asm = os.path.join(out_dir, src_name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", *defines, "--cuda-device-only", "-S", os.path.join(CSRC, src_name), "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
    return open(asm).read()


def resources(asm_text: str) -> list[dict]:
    """one dict per kernel of the code object's metadata: mangled name, vgpr (arch VGPRs incl. the AGPR part on the unified
    file), agpr, lds bytes, scratch bytes, sgpr, max workgroup size"""
    rows = []
    pat = (r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.max_flat_workgroup_size:\s+(\d+).*?\.name:\s+(\S+).*?"
           r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)")
    for m in re.finditer(pat, asm_text, re.S):
        ag, lds, wg, name, priv, sg, vg = m.groups()
        rows.append({"name": name, "vgpr": int(vg), "agpr": int(ag), "lds": int(lds), "scratch": int(priv), "sgpr": int(sg), "wg": int(wg)})
    return rows


CLASSES = {
    "mfma": re.compile(r"^v_mfma_"),
    "lds_read": re.compile(r"^ds_read|^ds_load"),
    "lds_write": re.compile(r"^ds_write|^ds_store"),
    "global_load": re.compile(r"^global_load|^buffer_load"),
    "global_store": re.compile(r"^global_store|^buffer_store"),
    "global_atomic": re.compile(r"^global_atomic|^buffer_atomic"),
    "barrier": re.compile(r"^s_barrier"),
    "scratch": re.compile(r"^scratch_"),
}


def kernel_lines(asm_text: str, mangled: str) -> list[str]:
    m = re.search(r"^" + re.escape(mangled) + r":[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M)
    if not m:
        raise KeyError(mangled)
    return [ln.split(";")[0].strip() for ln in m.group(1).split("\n") if ln.split(";")[0].strip()]


def count_classes(lines) -> dict:
    out = {k: 0 for k in CLASSES}
    for ln in lines:
        for k, rx in CLASSES.items():
            if rx.match(ln):
                out[k] += 1
    return out


def loop_profile(asm_text: str, mangled: str) -> dict:
    """Instruction mix of the kernel's STEADY-STATE hot loop and of the whole kernel.  The hot loop is the innermost
    (shortest) backward-branch loop that holds MFMAs AND global loads - the k-loop of the GEMM kernels, whose peeled last
    iterations (MFMAs, no loads) and enclosing tile loop are also loops of the listing; without such a loop, the
    shortest one that holds MFMAs.  {"loop": {...}, "total": {...}, "loop_lines": n}"""
    lines = kernel_lines(asm_text, mangled)
    label_at = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", ln)
        if not m:
            continue
        j = label_at.get(m.group(1) or m.group(2))
        if j is None or j >= i:
            continue  # forward branch
        body = lines[j:i + 1]
        loops.append((len(body), count_classes(body)))
    total = count_classes(lines)
    cands = [lp for lp in loops if lp[1]["mfma"] > 0 and lp[1]["global_load"] > 0] or [lp for lp in loops if lp[1]["mfma"] > 0]
    if not cands:
        return {"loop": {k: 0 for k in CLASSES}, "total": total, "loop_lines": 0}
    n, c = min(cands, key=lambda lp: lp[0])
    return {"loop": c, "total": total, "loop_lines": n}


def collect(tmp: str, experimental: bool = False) -> dict:
    """{source file: (asm text, [resource rows with 'pretty' names])} for every .hip of the product (experimental: of the
    library built with -DBGP_EXPERIMENTAL, battgp_amd/build.py --experimental)"""
    out = {}
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        text = compile_asm(src, tmp, ("-DBGP_EXPERIMENTAL",) if experimental else ())
        rows = resources(text)
        for r, dn in zip(rows, demangle([r["name"] for r in rows])):
            r["pretty"] = dn
        out[src] = (text, rows)
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        data = collect(tmp, experimental="--experimental" in sys.argv)
    print(f"{'kernel':78s} {'vgpr':>5} {'agpr':>5} {'LDS B':>7} {'scratch':>7}")
    for src, (text, rows) in data.items():
        for r in rows:
            print(f"{r['pretty'][:78]:78s} {r['vgpr']:5d} {r['agpr']:5d} {r['lds']:7d} {r['scratch']:7d}")
    if "--loops" in sys.argv:
        print("\nsteady-state k-loop per gemm_nt_kernel instantiation: mfma / lds_read / lds_write / global_load / barrier / lines")
        text, rows = data["bgp_linalg.hip"]
        for r in rows:
            if "gemm_nt_kernel" in r["pretty"]:
                p = loop_profile(text, r["name"])
                lp = p["loop"]
                print(f"  {r['pretty']:40s} {lp['mfma']:4d} {lp['lds_read']:4d} {lp['lds_write']:4d} {lp['global_load']:4d} {lp['barrier']:3d} {p['loop_lines']:5d}"
                      f"   | whole kernel: global_load {p['total']['global_load']}, global_store {p['total']['global_store']}, atomics {p['total']['global_atomic']}")


if __name__ == "__main__":
    sys.exit(main())
