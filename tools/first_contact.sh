#!/bin/bash
# The driver's literal round-end sequence on HEAD, in its order (VERDICT r3 "next round" item 1):
#   pytest -m gpu -x -q ; __graft_entry__.smoke() ; python3 bench.py --gpus 1 --steps 20 --warmup 5
#       gpurun --timeout 2400 -- 'bash tools/first_contact.sh r04a'
TAG=${1:-r06a}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
(time timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest_gpu.log; tail -3 $O/smoke.log; cut -c1-1500 $O/bench.json
