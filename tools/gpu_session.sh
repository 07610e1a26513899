#!/bin/bash
# One gpurun call that collects everything a round needs, most important first (a cut-off call loses the tail only):
#   1 the -m gpu suite   2 the default bench line   3 rocprofv3 kernel stats of the headline config
#   4 the PMC passes of the headline config (separate passes, tools/profile_round.sh)   5 mid-N sweep, system flow,
#   training iteration, steady fill, single-rank proxy of the sharded driver (plain and through a one-process RCCL group)
#       gpurun --timeout 3300 -- 'bash tools/gpu_session.sh r02 [stage ...]'
# Everything lands under gpurun_out/<tag>s/ ; copy what is to be judged into profiles/ afterwards
# (python tools/pmc_summary.py gpurun_out/<tag>s/pmc131k <tag> 131072 matern32).
TAG=${1:-r06}; shift
STAGES=${*:-"tests bench stats pmc sweep slim system train fill sharded optional"}
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG}s; mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STAGES " == *" $1 "* ]]; }
stamp() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT/session.log; }

if has tests; then
  stamp "pytest -m gpu"
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=10 --durations=15 > $OUT/pytest_gpu.log 2>&1
  stamp "pytest rc=$? $(tail -1 $OUT/pytest_gpu.log)"
fi
if has bench; then
  stamp "bench default"
  # bench.py's own wall budget (780 s) ends inside the stage's limit: the record is printed by bench.py, not by SIGTERM
  timeout 1000 python bench.py --steps 3 --warmup 1 --wall-budget-s 780 > $OUT/bench_default.json 2> $OUT/bench_default.err
  stamp "bench rc=$? $(cut -c1-200 $OUT/bench_default.json)"
fi
if has stats; then
  stamp "rocprofv3 kernel stats, N = 131072 matern32"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats131k -o st -- \
     python $REPO/bench.py --steps 1 --warmup 1 --no-extras --cpu-n 0 --no-residuals > $OUT/stats131k.log 2>&1)
  find $OUT/stats131k -name '*kernel_trace.csv' -delete   # hundreds of thousands of rows; the stats file is what is kept
  # the ORDERED-LAUNCH pass (look-ahead word 1|8: the panel stream's updates are ordered before rest(k), so the trailing
  # update's launches do not overlap): sum of gemm_nt_kernel<128,128,2> durations <= step time, which makes the roofline
  # fraction recomputable from the stats file alone, without an overlap argument
  stamp "rocprofv3 kernel stats, N = 131072 matern32, ordered launches (--lookahead 9)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats131k_ordered -o st -- \
     python $REPO/bench.py --steps 1 --warmup 1 --lookahead 9 --no-extras --cpu-n 0 --no-residuals > $OUT/stats131k_ordered.log 2>&1)
  find $OUT/stats131k_ordered -name '*kernel_trace.csv' -delete
fi
if has pmc; then
  stamp "PMC passes, N = 131072 matern32"
  bash tools/profile_round.sh ${TAG}s/pmc131k 131072 matern32 > $OUT/pmc131k.log 2>&1
  stamp "PMC passes, N = 40000 battgp"
  bash tools/profile_round.sh ${TAG}s/pmc40k 40000 battgp > $OUT/pmc40k.log 2>&1
fi
if has sweep; then
  stamp "sweep over N"
  BGP_ONLY=battgp timeout 600 python tools/sweep_n.py 5 1024 2048 4096 8192 16384 32768 40000 65536 > $OUT/sweep_n.jsonl 2> $OUT/sweep_n.err
fi
if has slim; then
  stamp "slim chain kernels (+32), split panels (+64), fused update + tile Cholesky (+128): A/B, scheme 1"
  export BGP_EXPERIMENTAL_LIB=1   # the optional families live in libbattgp_exp.so only (battgp_amd/build.py --experimental)
  for la in 1 33 65 97 129 193; do
    BGP_LA=$la BGP_SCHEME=1 BGP_ONLY=battgp timeout 600 python tools/sweep_n.py 5 4096 8192 16384 24576 32768 40000 65536 > $OUT/sweep_la$la.jsonl 2>> $OUT/sweep_n.err
  done
  timeout 600 python tools/ab_lookahead.py 3 4096 16384 40000 > $OUT/ab_lookahead.jsonl 2>> $OUT/sweep_n.err   # with the bit-identity flag
  for la in 1 129; do BGP_LA=$la BGP_ONLY=battgp timeout 300 python tools/sweep_n.py 9 1024 2048 4096 8192 > $OUT/sweep_small_la$la.jsonl 2>> $OUT/sweep_n.err; done
  (cd /tmp && BGP_LA=97 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl16k_slim -o tl -- \
     python $REPO/tools/profile_workload.py 16384 battgp 3 > $OUT/tl16k_slim.log 2>&1)
  python tools/timeline.py $(find $OUT/tl16k_slim -name '*kernel_trace.csv' | head -1) > $OUT/tl16k_slim_summary.txt 2>&1
  gzip -f $(find $OUT/tl16k_slim -name '*kernel_trace.csv') 2>/dev/null
  unset BGP_EXPERIMENTAL_LIB
fi
if has system; then
  stamp "system flow (1 pack + 8 cells)"
  timeout 600 python tools/system_probe.py 1000 4000 16000 > $OUT/system_auto.jsonl 2> $OUT/system.err
  BGP_STREAMS=1 timeout 600 python tools/system_probe.py 1000 4000 16000 > $OUT/system_seq.jsonl 2>> $OUT/system.err
fi
if has train; then
  stamp "optimiser iteration (LML + gradient)"
  timeout 600 python tools/train_iter.py 1000 4000 16000 40000 > $OUT/train_iter.jsonl 2> $OUT/train_iter.err
  # the in-place-inverse gradient where the second N^2 buffer of round 2 would no longer have fitted next to the factor
  timeout 600 python tools/grad_probe.py 131072 > $OUT/grad_probe_131072.txt 2>&1
fi
if has fill; then
  stamp "steady fill"
  timeout 300 python tools/fill_rate.py 131072 > $OUT/fill_rate.txt 2>&1
  BGP_EXPERIMENTAL_LIB=1 BGP_FILL_TABLE=256 timeout 300 python tools/fill_rate.py 131072 > $OUT/fill_rate_table256.txt 2>&1
  BGP_EXPERIMENTAL_LIB=1 BGP_FILL_MFMA=1 timeout 300 python tools/fill_rate.py 131072 > $OUT/fill_rate_mfma.txt 2>&1
  BGP_EXPERIMENTAL_LIB=1 BGP_FILL_MFMA=1 BGP_FILL_TABLE=256 timeout 300 python tools/fill_rate.py 131072 > $OUT/fill_rate_mfma_table256.txt 2>&1
fi
if has sharded; then
  stamp "sharded driver, single-rank proxy"
  timeout 600 python bench.py --mode sharded --n 65536 --kernel battgp --steps 1 --warmup 1 --cpu-n 0 --no-extras --sharded-grad > $OUT/sharded_n65536.json 2> $OUT/sharded.err
  timeout 600 python bench.py --mode sharded --n 65536 --kernel battgp --steps 1 --warmup 1 --cpu-n 0 --no-extras --sharded-grad --force-group > $OUT/sharded_n65536_rccl.json 2>> $OUT/sharded.err
  timeout 900 python bench.py --mode sharded --n 131072 --kernel battgp --steps 1 --warmup 0 --cpu-n 0 --no-extras > $OUT/sharded_n131072.json 2>> $OUT/sharded.err
fi
if has optional; then
  stamp "optional schedules / fill variants through the test suite (child processes; a failure is a failure)"
  BGP_TEST_OPTIONAL=1 BGP_EXPERIMENTAL_LIB=1 timeout 1500 python -m pytest tests/test_gpu_zz_optional_schedules.py -m gpu -q -rfE > $OUT/pytest_optional.log 2>&1
  stamp "optional rc=$? $(tail -1 $OUT/pytest_optional.log)"
fi
python tools/decide_ab.py $OUT > $OUT/ab_decision.txt 2>&1   # the promote / delete list of DESIGN.md section 10.2, from the files above
stamp done
ls -la $OUT
