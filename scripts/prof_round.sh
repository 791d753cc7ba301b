#!/bin/bash
# One GPU call -> everything profiles/<tag>_* is built from (gpurun_out/<tag>/):
#   trace/  rocprofv3 --kernel-trace of bench.py (timed region + evaluation, gather, Coach and sampler legs; no aten baselines, no child legs)
#   fetch/, write/  the two --pmc passes of scripts/pmc_step.py        c5fetch/, c5write/  the same two passes of scripts/pmc_c5.py
#   pmc1/, pmc2/    two passes of 7 SQ counters of scripts/pmc_step.py
#   usage (on the GPU box, from the repo root): bash scripts/prof_round.sh <tag>   the summaries land in gpurun_out/<tag>/profiles/ (copy them into profiles/)
tag=$1
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out/trace $out/fetch $out/write $out/c5fetch $out/c5write $out/pmc1 $out/pmc2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-baselines --no-c5 --no-legs > $out/bench_traced.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o pmc -- python3 $root/scripts/pmc_step.py > $out/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o pmc -- python3 $root/scripts/pmc_step.py > $out/write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out/pmc1 -o pmc -- python3 $root/scripts/pmc_step.py > $out/pmc1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/pmc2 -o pmc -- python3 $root/scripts/pmc_step.py > $out/pmc2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/c5fetch -o pmc -- python3 $root/scripts/pmc_c5.py 20 > $out/c5fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/c5write -o pmc -- python3 $root/scripts/pmc_c5.py 20 > $out/c5write.log 2>&1 &&
cd $root && timeout -k 10 600 python3 bench.py > $out/bench.log 2>&1 &&
grep '^{"metric"' $out/bench.log > $out/bench.json &&
python3 scripts/kstats.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 221 24 | tee $out/kstats.txt &&
python3 scripts/make_profiles.py $out $tag $out/profiles | tee $out/make_profiles.txt
# the raw databases are far beyond what travels back (64 MiB): only the summaries and logs stay
rm -rf $out/trace $out/fetch $out/write $out/c5fetch $out/c5write $out/pmc1 $out/pmc2
