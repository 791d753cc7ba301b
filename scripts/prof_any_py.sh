#!/bin/bash
# kernel trace of any python script: bash scripts/prof_any_py.sh <tag> <steps for the per-step table> script.py args...  -> gpurun_out/<tag>_kstats.txt
tag=$1; steps=$2; shift 2
root=$PWD; out=$root/gpurun_out/trace_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 "$root/$1" "${@:2}" > $out/log.txt 2>&1
cd $root
python3 scripts/kstats.py $(ls $out/*/*.db $out/*.db 2>/dev/null | head -1) $steps 16 > gpurun_out/${tag}_kstats.txt
rm -rf $out
cat gpurun_out/${tag}_kstats.txt
