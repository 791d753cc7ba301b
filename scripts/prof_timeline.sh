#!/bin/bash
root=$PWD; out=$root/gpurun_out/tl; mkdir -p $out/trace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 $root/bench.py --steps 200 --warmup 20 --no-extras $TL_BENCH_ARGS > $out/log.txt 2>&1
cd $root
python3 scripts/step_timeline.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 20 | tee $out/timeline.txt
python3 scripts/step_timeline.py $(ls $out/trace/*/*.db $out/trace/*.db 2>/dev/null | head -1) 21 >> $out/timeline.txt
rm -rf $out/trace
