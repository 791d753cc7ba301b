#!/bin/bash
# One GPU call -> everything profiles/<tag>_* is built from (gpurun_out/<tag>/):
#   trace/  rocprofv3 --kernel-trace of bench.py's timed region       fetch/, write/  the two --pmc passes of scripts/pmc_step.py
#   usage (on the GPU box, from the repo root): bash scripts/prof_round.sh <tag>   then here: python scripts/make_profiles.py gpurun_out/<tag> <tag>
tag=$1
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out/trace $out/fetch $out/write
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-extras > $out/bench_traced.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o pmc -- python3 $root/scripts/pmc_step.py > $out/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o pmc -- python3 $root/scripts/pmc_step.py > $out/write.log 2>&1 &&
cd $root && timeout -k 10 400 python3 bench.py > $out/bench.log 2>&1 &&
grep '^{"metric"' $out/bench.log > $out/bench.json &&
python3 scripts/kstats.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 221 20 | tee $out/kstats.txt
