#!/bin/bash
# timeline of one steady-state step of a bench leg: bash scripts/prof_leg_timeline.sh config5  -> gpurun_out/legtl_<name>/timeline.txt
leg=$1
root=$PWD
out=$root/gpurun_out/legtl_$leg
mkdir -p $out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 $root/bench_legs.py --leg $leg > $out/log.txt 2>&1
cd $root
python3 scripts/step_timeline.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 20 | tee $out/timeline.txt
rm -rf $out/trace
