#!/bin/bash
# SQ counters of the step's kernels (two rocprofv3 --pmc passes of scripts/pmc_step.py) -> gpurun_out/<tag>/pmc{1,2}; then
#   python scripts/pmc_kernel.py gpurun_out/<tag>/pmc1 enc_step_k   (etc.)
tag=$1
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out/pmc1 $out/pmc2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $out/pmc1 -o pmc -- python3 $root/scripts/pmc_step.py > $out/pmc1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $out/pmc2 -o pmc -- python3 $root/scripts/pmc_step.py > $out/pmc2.log 2>&1
cd $root
for k in enc_step_k scatter_owner_k enc_wgrad_k sasrec_batch_prep_k; do echo "== $k"; python3 scripts/pmc_kernel.py $out/pmc1 $k; python3 scripts/pmc_kernel.py $out/pmc2 $k; done
