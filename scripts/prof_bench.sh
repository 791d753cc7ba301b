#!/bin/bash
# rocprofv3 kernel trace of the bench's timed region -> per-step kernel table (gpurun_out/<tag>/)
#   usage (on the GPU box, from the repo root): bash scripts/prof_bench.sh <tag> [bench.py args]
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-extras "$@" > $out/bench.log 2>&1
grep '^{"metric"' $out/bench.log > $out/bench.json
python3 $root/scripts/kstats.py $(ls $out/*.db | head -1) 221 > $out/kstats.txt
cat $out/kstats.txt
