"""Static resource table of the shipped kernels (VGPR / AGPR / LDS / scratch per kernel, from the gfx950 assembly's
metadata): what decides how many workgroups of which kernels can share a CU.   python tools/kernel_resources.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "battgp_amd", "csrc")


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
            return [re.sub(r"^void ", "", re.sub(r"\(.*", "", o.replace("(anonymous namespace)::", ""))) for o in out]
        except (FileNotFoundError, subprocess.CalledProcessError):
            continue
    return names


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            asm = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                            os.path.join(CSRC, src), "-o", asm], check=True, stderr=subprocess.DEVNULL)
            text = open(asm).read()
            pat = (r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                   r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)")
            for m in re.finditer(pat, text, re.S):
                ag, lds, name, priv, sg, vg = m.groups()
                rows.append((src, name, int(vg), int(ag), int(lds), int(priv)))
    names = demangle([r[1] for r in rows])
    print(f"{'kernel':78s} {'vgpr':>5} {'agpr':>5} {'LDS B':>7} {'scratch':>7}")
    for (src, _, vg, ag, lds, priv), dn in zip(rows, names):
        print(f"{dn[:78]:78s} {vg:5d} {ag:5d} {lds:7d} {priv:7d}")


if __name__ == "__main__":
    sys.exit(main())
