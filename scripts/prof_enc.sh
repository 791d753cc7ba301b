#!/bin/bash
# rocprofv3 kernel durations of the encoder launches per sequence-length class (event timing is CPU-launch-bound at these sizes)
root=$PWD
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  out=$root/gpurun_out/pe_$k
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace -d $out -o trace -- python3 $root/scripts/enc_time.py --kinds $k --B ${ENC_B:-512} > $out/log.txt 2>&1
  echo "== $k: $(grep items $out/log.txt | cut -c1-60)"
  python3 - $out <<'PY'
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
for n, c, a, mn in db.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name like '%enc_%' or name like '%batch_prep%' group by name order by 3 desc"):
    print(f"   avg {a/1e3:7.1f} us  min {mn/1e3:7.1f}  calls {c:4d}  {n[:50]}")
PY
done
