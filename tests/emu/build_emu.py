"""TEST INFRASTRUCTURE ONLY: builds tests/emu/_build/libbattgp_emu.so - the kernel and host SOURCES of
battgp_amd/csrc compiled for this container's CPU against tests/emu/hip/hip_runtime.h (cooperative-fiber
execution of the workgroups, see hipemu.cpp).  Used by tests/test_emu_kernels.py to check the kernels' indexing and
algebra against the oracle while no GPU is at hand, and as the sanitizer target (``sanitize=True``).
The product binding (battgp_amd/_lib.py) never looks here."""

from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "battgp_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
SOURCES = [os.path.join(CSRC, f) for f in ("bgp_fill.hip", "bgp_linalg.hip", "bgp_capi.hip")] + [os.path.join(HERE, "hipemu.cpp")]


def _compiler() -> str:
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "clang++"):
        if os.path.exists(c) or c == "clang++":
            return c
    return "clang++"


def sanitizer_runtime(kind: str = "asan") -> str:
    """Path of the shared sanitizer runtime (to LD_PRELOAD into a python that loads the sanitized library)."""
    name = {"asan": "libclang_rt.asan-x86_64.so", "ubsan": "libclang_rt.ubsan_standalone-x86_64.so"}[kind]
    return subprocess.run([_compiler(), f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()


def experimental_default() -> bool:
    """BGP_EMU_EXPERIMENTAL=1: the CPU build of the EXPERIMENTAL library (-DBGP_EXPERIMENTAL: the optional kernel families);
    otherwise the CPU build is the default library's sources, like the product"""
    return os.environ.get("BGP_EMU_EXPERIMENTAL") == "1"


def build(sanitize=False, force: bool = False, verbose: bool = False, experimental=None) -> str:
    """``sanitize``: False, True / "ubsan" (UndefinedBehaviorSanitizer) or "asan" (AddressSanitizer; the fiber
    switches are announced to it, LD_PRELOAD ``sanitizer_runtime("asan")`` into the loading process).
    ``experimental``: compile with -DBGP_EXPERIMENTAL (None: as BGP_EMU_EXPERIMENTAL says)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    if experimental is None:
        experimental = experimental_default()
    kind = "asan" if sanitize == "asan" else ("ubsan" if sanitize else "")
    tag = ("exp" if experimental else "") + (("_" if experimental and kind else "") + kind)
    lib = os.path.join(OUT_DIR, f"libbattgp_emu_{tag}.so" if tag else "libbattgp_emu.so")
    kind_obj = tag  # object files per configuration
    deps = SOURCES + [os.path.abspath(__file__), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(CSRC, "bgp_internal.h"),
                      os.path.join(CSRC, "bgp_fill_tile.inc"), os.path.join(ROOT, "include", "battgp.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in deps):
        return lib
    objs = []
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-march=native", "-ffp-contract=fast", "-pthread", f"-I{HERE}", "-Wall",
             "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable", "-Wno-unused-value"]
    if experimental:
        flags.append("-DBGP_EXPERIMENTAL")
    if kind == "ubsan":
        flags += ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    elif kind == "asan":
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer"]
        flags[flags.index("-O2")] = "-O1"
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = os.path.join(OUT_DIR, os.path.basename(src) + (f".{kind_obj}.o" if kind_obj else ".o"))
        cmd = [_compiler(), "-x", "c++", *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as pool:  # the translation units side by side
        objs = list(pool.map(compile_one, SOURCES))
    # thread-local markers linked around the two kernel objects: what lies between them in a thread's TLS block is the
    # kernels' LDS (their `static thread_local` arrays), which HIPEMU_POISON=ff fills before every workgroup (hipemu.cpp)
    marks = []
    for name in ("hipemu_lds_begin", "hipemu_lds_end"):
        msrc, mobj = os.path.join(OUT_DIR, name + ".cpp"), os.path.join(OUT_DIR, name + (f".{kind_obj}.o" if kind_obj else ".o"))
        with open(msrc, "w") as f:
            f.write(f"thread_local char {name}[64];\n")
        subprocess.run([_compiler(), "-std=c++17", "-O1", "-fPIC", "-c", msrc, "-o", mobj], check=True)
        marks.append(mobj)
    kernels = [o for o in objs if "bgp_fill" in o or "bgp_linalg" in o]
    objs = [marks[0], *kernels, marks[1], *[o for o in objs if o not in kernels]]
    san_link = []
    if kind:  # the sanitizer runtime as a shared object next to the compiler, found through an rpath
        rt_dir = os.path.dirname(sanitizer_runtime(kind))
        san_link = ["-fsanitize=address" if kind == "asan" else "-fsanitize=undefined", "-shared-libsan", f"-Wl,-rpath,{rt_dir}"]
    cmd = [_compiler(), "-shared", "-pthread", *san_link, *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    import sys

    print(build(sanitize="asan" if "--asan" in sys.argv else ("--sanitize" in sys.argv), force=True, verbose=True,
                experimental=True if "--experimental" in sys.argv else None))
