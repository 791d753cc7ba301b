#!/bin/bash
# kernel statistics of the sharded config-5 step on ONE rank (RCCL world of 1): bash scripts/prof_c5_sharded.sh -> gpurun_out/c5s/kstats.txt
root=$PWD; out=$root/gpurun_out/c5s; mkdir -p $out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/scripts/c5_sharded_world1.py > $out/log.txt 2>&1
cd $root
python3 scripts/kstats.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 24 30 > $out/kstats.txt
rm -rf $out/trace
tail -3 $out/log.txt | cut -c1-400; head -34 $out/kstats.txt
