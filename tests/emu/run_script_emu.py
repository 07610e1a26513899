"""TEST INFRASTRUCTURE ONLY: runs one of the repository's GPU scripts (bench.py, tools/*.py) with the CPU build of the
kernel sources behind the binding and torch's CUDA entry points faked (inject.py), so that their CONTROL FLOW - argument
parsing, torch.distributed.run rendezvous, barriers, max-over-ranks timing, the per-rank cell workload, the sharded
sub-record, the JSON lines - can be rehearsed without a GPU (tests/test_bench_host.py).  Nothing they print under this
wrapper is a measurement.

    python tests/emu/run_script_emu.py bench.py --steps 1 --warmup 0 --size 600 --no-extras
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tests/emu/run_script_emu.py bench.py --gpus 2 --steps 1 --warmup 0 --size 600 --backend gloo --sharded-n 384 --sharded-nb 128
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from inject import fake_cuda_tensors, installed  # noqa: E402

fake_cuda_tensors()
with installed():
    script = os.path.join(ROOT, sys.argv[1])
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")
