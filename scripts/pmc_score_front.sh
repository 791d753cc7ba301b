#!/bin/bash
# SQ counters of score_front_k (20 re_score_topk calls, scripts/x2_prof.py): bash scripts/pmc_score_front.sh [kernel substring]
root=$PWD; out=$root/gpurun_out/pmc_front; rm -rf $out; mkdir -p $out
k=${1:-score_front_k}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $out/p1 -o pmc -- python3 $root/scripts/x2_prof.py > $out/log1.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out/p2 -o pmc -- python3 $root/scripts/x2_prof.py > $out/log2.txt 2>&1
cd $root
(python3 scripts/pmc_kernel.py $out/p1 $k; python3 scripts/pmc_kernel.py $out/p2 $k) > $out/summary.txt 2>&1
rm -rf $out/p1 $out/p2
cat $out/summary.txt
