#!/bin/bash
# rocprofv3 kernel trace of a python script; prints the dispatches of kernels matching <pattern> in launch order
#   usage: bash scripts/prof_seq.sh <tag> <pattern> script.py [args]
tag=$1; pat=$2; shift; shift
root=$PWD; out=$root/gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o trace -- python3 "$@" > $out/log.txt 2>&1
python3 - $out "$pat" <<'PY'
import sqlite3, sys, glob
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
for n, s, e in db.execute("select name, start, end from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)):
    print(f"{(e - s) / 1e3:8.1f} us  {n[:50]}")
PY
