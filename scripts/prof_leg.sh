#!/bin/bash
# kernel trace of one bench leg: bash scripts/prof_leg.sh config4  -> gpurun_out/leg_<name>/kstats.txt
leg=$1
root=$PWD
out=$root/gpurun_out/leg_$leg
mkdir -p $out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench_legs.py --leg $leg > $out/log.txt 2>&1
cd $root
python3 scripts/kstats.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 1 40 > $out/kstats.txt
rm -rf $out/trace
tail -45 $out/kstats.txt
