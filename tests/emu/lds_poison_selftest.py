"""TEST INFRASTRUCTURE ONLY: self-test of the LDS poison of the CPU build (hipemu.cpp::poison_lds, HIPEMU_POISON=ff).  A probe
kernel that READS a __shared__ array it never wrote is compiled against the stand-in <hip/hip_runtime.h> and linked the way
build_emu.py links the real kernel objects (between the two thread-local markers): without the switch the array reads as
zeros (what made the CPU build kinder than the GPU), with it every element is a NaN - i.e. a kernel of the product that
depended on unwritten LDS could not pass the poisoned runs.      python tests/emu/lds_poison_selftest.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

PROBE = r'''
#include <hip/hip_runtime.h>
__global__ void probe_kernel(double* out, int never) {
  __shared__ double t[64];
  __shared__ int flag;
  if (never) {  // (a store the compiler cannot rule out: otherwise it folds the loads of a never-written static to 0)
    t[threadIdx.x] = 1.0;
    flag = 7;
  }
  out[threadIdx.x] = t[threadIdx.x];
  if (threadIdx.x == 0) out[64] = (double)flag;
}
extern "C" void run_probe(double* out) {
  hipLaunchKernelGGL(probe_kernel, dim3(3), dim3(64), 0, nullptr, out, (int)(out[64] > 1e300));
  hipDeviceSynchronize();
}
'''


def main():
    cxx = build_emu._compiler()
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "probe.cpp")
        with open(src, "w") as f:
            f.write(PROBE)
        objs = {}
        for name, text in (("hipemu_lds_begin", "thread_local char hipemu_lds_begin[64];\n"), ("hipemu_lds_end", "thread_local char hipemu_lds_end[64];\n")):
            p = os.path.join(d, name + ".cpp")
            with open(p, "w") as f:
                f.write(text)
            objs[name] = os.path.join(d, name + ".o")
            subprocess.run([cxx, "-std=c++17", "-O1", "-fPIC", "-c", p, "-o", objs[name]], check=True)
        flags = ["-std=c++17", "-O2", "-fPIC", "-pthread", f"-I{HERE}", "-Wno-unknown-attributes"]
        subprocess.run([cxx, "-x", "c++", *flags, "-c", src, "-o", os.path.join(d, "probe.o")], check=True)
        subprocess.run([cxx, "-x", "c++", *flags, "-c", os.path.join(HERE, "hipemu.cpp"), "-o", os.path.join(d, "hipemu.o")], check=True)
        lib = os.path.join(d, "libprobe.so")
        subprocess.run([cxx, "-shared", "-pthread", objs["hipemu_lds_begin"], os.path.join(d, "probe.o"), objs["hipemu_lds_end"],
                        os.path.join(d, "hipemu.o"), "-o", lib], check=True)
        code = ("import ctypes as C, sys; lib = C.CDLL(sys.argv[1]); out = (C.c_double * 65)(); lib.run_probe(out); "
                "v = list(out); print('nan' if all(x != x for x in v[:64]) else ('zero' if all(x == 0.0 for x in v[:64]) else 'mixed'), int(v[64]) if v[64] == v[64] else 'nan')")
        res = {}
        for tag, env in (("plain", {}), ("poison", {"HIPEMU_POISON": "ff"})):
            r = subprocess.run([sys.executable, "-c", code, lib], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            res[tag] = r.stdout.split()
        print(res)
        assert res["plain"] == ["zero", "0"], res     # unwritten LDS reads as zeros in the plain CPU build ...
        assert res["poison"] == ["nan", "-1"], res    # ... and as NaN / -1 under the switch
        print("ok: the LDS poison reaches unwritten __shared__ arrays")


if __name__ == "__main__":
    main()
