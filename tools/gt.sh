#!/bin/bash
# tools/gt.sh <logname> <pytest args...>: GPU pytest on the box, log under gpurun_out/r2/, failures summarised
mkdir -p gpurun_out/r2
log=gpurun_out/r2/$1.log; shift
timeout -k 10 1000 python -m pytest "$@" -q -m gpu > $log 2>&1
rc=$?
grep -n "^E  \|^>\|^FAILED\|^ERROR" $log | head -40
tail -3 $log
exit $rc
