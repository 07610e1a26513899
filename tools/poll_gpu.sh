#!/bin/bash
# Keeps asking gpurun for a box (a refused call costs nothing) and runs ONE command the first time the pool accepts it.
#     bash tools/poll_gpu.sh <marker-file> <gpurun-timeout-s> '<command>'
# Exit code 2 = refused / closed, 3 = no box free: both are retried every 5 minutes.  Anything else = the call ran:
# the marker file gets gpurun's exit code and the loop ends (gpurun_out/.last_call.json holds the verdict).
MARK=$1; LIMIT=$2; CMD=$3
rm -f "$MARK"
while true; do
  /usr/local/graft/bin/gpurun --timeout "$LIMIT" -- "$CMD" > "${MARK}.out" 2>&1
  rc=$?
  if [ $rc -ne 2 ] && [ $rc -ne 3 ]; then echo "$rc $(date +%H:%M:%S)" > "$MARK"; exit 0; fi
  echo "$(date +%H:%M:%S) rc=$rc" >> "${MARK}.poll"
  sleep 300
done
