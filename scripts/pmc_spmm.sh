#!/bin/bash
# L2 hit rate of one LightGCN propagation (re_spmm_csr / re_spmm_csr_split) on the Yelp2018-shaped graph: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum
# over scripts/spmm_split_ab.py <variant>.    bash scripts/pmc_spmm.sh plain "split auto"   -> gpurun_out/pmc_spmm/summary.txt
root=$PWD; out=$root/gpurun_out/pmc_spmm; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/p$i -o pmc -- python3 $root/scripts/spmm_split_ab.py "$v" > $out/run$i.log 2>&1
  (echo "== $v"; cd $root; python3 scripts/pmc_kernel.py $out/p$i spmm_csr_rows) >> $out/summary.txt 2>&1
  rm -rf $out/p$i
done
cd $root
cat $out/summary.txt
