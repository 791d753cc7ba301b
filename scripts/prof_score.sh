#!/bin/bash
# kernel trace of 20 re_score_topk calls (Beauty shape, iid scores): bash scripts/prof_score.sh -> gpurun_out/score_kstats.txt
root=$PWD; out=$root/gpurun_out/score_trace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $root/scripts/x2_prof.py > $out/log.txt 2>&1
cd $root
python3 scripts/kstats.py $(ls $out/*/*.db $out/*.db 2>/dev/null | head -1) 20 20 > gpurun_out/score_kstats.txt
rm -rf $out
cat gpurun_out/score_kstats.txt
