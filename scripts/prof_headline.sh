#!/bin/bash
# kernel trace of the headline step only: bash scripts/prof_headline.sh -> gpurun_out/headline_kstats.txt
root=$PWD; out=$root/gpurun_out/headline_trace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-legs --no-c5 --no-cpu-baseline --no-baselines --no-extras > $out/log.txt 2>&1
cd $root
python3 scripts/kstats.py $(ls $out/*/*.db $out/*.db 2>/dev/null | head -1) 221 16 > gpurun_out/headline_kstats.txt
tail -3 $out/log.txt | cut -c1-300
rm -rf $out
cat gpurun_out/headline_kstats.txt
