#!/bin/bash
# The driver's round-end sequence on HEAD, in its order (VERDICT r3 "next round" item 1; pytest without -x, see below):
#   pytest -m gpu -x -q ; __graft_entry__.smoke() ; python3 bench.py --gpus 1 --steps 20 --warmup 5
#       gpurun --timeout 3900 -- 'bash tools/first_contact.sh r06a'
TAG=${1:-r06a}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
# the driver stops at the first failure (-x); a builder's lease wants every failure of ONE run on record (GPU minutes are
# scarce): same order (tests/conftest.py GPU_ORDER), up to 12 failures, reasons of skips and failures listed
(time timeout 2100 python -m pytest tests -m gpu --maxfail=12 -q -rfEs --durations=25) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
# second argument "short": a lease accepted late in a session - 3 timed steps and a 7-minute wall budget instead of the driver's 20 + 5
if [ "${2:-}" = short ]; then BENCH_ARGS="--steps 3 --warmup 1 --wall-budget-s 420"; else BENCH_ARGS="--steps 20 --warmup 5"; fi
(time timeout 1700 python3 bench.py --gpus 1 $BENCH_ARGS) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest_gpu.log; tail -3 $O/smoke.log; cut -c1-1500 $O/bench.json
