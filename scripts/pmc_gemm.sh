#!/bin/bash
# Where the wide fp32 product's cycles go (two rocprofv3 --pmc passes over scripts/gemm_one.py): MFMA busy / wave cycles / waits, LDS conflicts.
#   bash scripts/pmc_gemm.sh [M N K tA tB]   -> gpurun_out/pmc_gemm/summary.txt
root=$PWD; out=$root/gpurun_out/pmc_gemm; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $out/p1 -o pmc -- python3 $root/scripts/gemm_one.py "$@" > $out/log1.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $out/p2 -o pmc -- python3 $root/scripts/gemm_one.py "$@" > $out/log2.txt 2>&1
cd $root
(python3 scripts/pmc_kernel.py $out/p1 gemm_wide; python3 scripts/pmc_kernel.py $out/p2 gemm_wide) > $out/summary.txt 2>&1
rm -rf $out/p1 $out/p2
cat $out/summary.txt
