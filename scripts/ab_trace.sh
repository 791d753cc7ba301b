#!/bin/bash
# kernel traces of scripts/lb_one.py in the _old worktree and here: bash scripts/ab_trace.sh 4096
root=$PWD; B=${1:-4096}
for side in _old .; do
  out=$root/gpurun_out/ab_$(basename $(realpath $side)); rm -rf $out; mkdir -p $out
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 $root/$side/scripts/lb_one.py $B > $out/log.txt 2>&1
  cd $root
  echo "== $side"; python3 scripts/kstats.py $(ls $out/*/*.db $out/*.db 2>/dev/null | head -1) 40 10
  rm -rf $out
done
