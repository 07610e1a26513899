"""Did a source change move the machine code of a kernel?  Compiles battgp_amd/csrc/<file>.hip of a git revision and of
the working tree for gfx950 and compares every kernel's instruction stream (labels neutralised, comments dropped).
Used to keep the kernels whose timings are on record byte-identical while new variants are added next to them.

    python tools/isa_diff.py 24155f9 [bgp_linalg.hip] [-DBGP_EXPERIMENTAL ...]   -> one line per kernel that differs / is new / is gone
Extra -D flags apply to the WORKING TREE's compile only (e.g. the experimental library against a revision from before the
switch existed)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(asm_path):
    text = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        body = [ln.split(";")[0].strip() for ln in m.group(2).split("\n")]
        body = [re.sub(r"\.LBB\d+_\d+", "L", ln) for ln in body if ln and not ln.startswith(".") and not ln.endswith(":")]
        out[m.group(1)] = (hashlib.md5("\n".join(body).encode()).hexdigest()[:12], len(body))
    return out


def compile_to_asm(src, inc_root, out, defines=()):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *defines, "--cuda-device-only", "-S", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL, cwd=inc_root)


def main():
    rev = sys.argv[1]
    defines = [a for a in sys.argv[2:] if a.startswith("-D")]
    rest = [a for a in sys.argv[2:] if not a.startswith("-D")]
    name = rest[0] if rest else "bgp_linalg.hip"
    with tempfile.TemporaryDirectory() as tmp:
        # the revision's sources in the same relative layout (csrc/ includes ../../include/battgp.h)
        for rel in ("battgp_amd/csrc/" + name, "battgp_amd/csrc/bgp_internal.h", "battgp_amd/csrc/bgp_fill_tile.inc", "include/battgp.h"):
            dst = os.path.join(tmp, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            got = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{rel}"], capture_output=True, text=True)
            if got.returncode != 0 and rel.endswith(".inc"):
                continue  # revisions from before the tile body was split out
            got.check_returncode()
            with open(dst, "w") as f:
                f.write(got.stdout)
        old_s, new_s = os.path.join(tmp, "old.s"), os.path.join(tmp, "new.s")
        compile_to_asm(os.path.join(tmp, "battgp_amd", "csrc", name), tmp, old_s)
        compile_to_asm(os.path.join(ROOT, "battgp_amd", "csrc", name), ROOT, new_s, defines)
        a, b = kernels(old_s), kernels(new_s)
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(set(a) | set(b))), capture_output=True, text=True).stdout.split("\n")
    names = dict(zip(sorted(set(a) | set(b)), (re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "").replace("void ", "")) for d in dem)))
    same = 0
    for k in sorted(set(a) | set(b)):
        if a.get(k) == b.get(k):
            same += 1
            continue
        what = "NEW " if k not in a else ("GONE" if k not in b else "MOVED")
        print(f"{what:5s} {names[k][:70]:70s} {a.get(k)} -> {b.get(k)}")
    print(f"{same} kernels with identical instruction streams in {rev} and the working tree ({name})")


if __name__ == "__main__":
    main()
