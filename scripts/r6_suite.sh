#!/bin/bash
# One full `-m gpu` suite run as the driver runs it, its summary line appended to gpurun_out/r6_suite_runs.txt (-> profiles/r6_gpu_suite_runs.txt).
# usage: scripts/r6_suite.sh <tag> [extra pytest args]
tag=$1; shift
log=gpurun_out/${tag}_full.log
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider "$@" > $log 2>&1
rc=$?
echo "rc $rc" >> $log
echo "$tag $(date -u +%FT%TZ) head=$(cat gpurun_out/.head 2>/dev/null) rc=$rc :: $(grep -E 'passed|failed|error' $log | tail -1)" >> gpurun_out/r6_suite_runs.txt
grep -a "capture-guard\|Memory access fault\|Aborted\|terminate called" $log | head -5
tail -3 $log
exit $rc
