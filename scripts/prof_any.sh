#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script; prints every dispatch's duration grouped by (kernel, grid)
tag=$1; shift
root=$PWD; out=$root/gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o trace -- python3 "$@" > $out/log.txt 2>&1
python3 - $out <<'PY'
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
g = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
q = f"select name, {g if g else 0}, count(*), avg(end-start), min(end-start) from kernels group by name, {g if g else 0} order by 1, 2"
for n, gx, c, a, mn in db.execute(q):
    if "at::native" in n or "rocclr" in n: continue
    print(f"   grid {gx:8}  calls {c:4d}  avg {a/1e3:7.1f} us  min {mn/1e3:7.1f}  {n[:60]}")
PY
