"""Build libbattgp.so (HIP, gfx950 only) in-tree with hipcc.  No fallback of any kind.

    python -m battgp_amd.build [--force] [--experimental]

``--experimental`` builds a SECOND library, libbattgp_exp.so, from the same sources with -DBGP_EXPERIMENTAL: the default
library plus the optional kernel families no MI355X has timed yet (slim / split / fused panel chain, table-256 and
matrix-pipe fill interiors; bgp_internal.h).  The product never loads it unless BGP_EXPERIMENTAL_LIB=1 is set
(battgp_amd/_lib.py): it exists for the A/B stage of tools/gpu_session.sh and the optional parity cases."""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbattgp.so")
LIB_EXPERIMENTAL = os.path.join(HERE, "libbattgp_exp.so")
SOURCES = ["bgp_fill.hip", "bgp_linalg.hip", "bgp_capi.hip"]
HEADERS = [os.path.join(CSRC, "bgp_internal.h"), os.path.join(CSRC, "bgp_fill_tile.inc"), os.path.join(os.path.dirname(HERE), "include", "battgp.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libbattgp.so cannot be built")


def needs_build(experimental: bool = False) -> bool:
    lib = LIB_EXPERIMENTAL if experimental else LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True, experimental: bool = False) -> str:
    lib = LIB_EXPERIMENTAL if experimental else LIB
    if not force and not needs_build(experimental):
        return lib
    hipcc = _hipcc()
    flags = FLAGS + (["-DBGP_EXPERIMENTAL"] if experimental else [])
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".exp.o" if experimental else ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as pool:  # the three translation units side by side
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv, experimental="--experimental" in sys.argv)
