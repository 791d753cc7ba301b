#!/bin/bash
# LDS bank-conflict cycles per LDS instruction of the step's kernels (one rocprofv3 --pmc pass of scripts/pmc_step.py)
root=$PWD; out=$root/gpurun_out/${1:-lds}; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $out -o pmc -- python3 $root/scripts/pmc_step.py > $out/log.txt 2>&1
python3 - $out <<'PY'
import sqlite3, sys, glob, collections
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if "pmc_event" in t][0]; info = [t for t in tabs if "info_pmc" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]; kd = [t for t in tabs if "kernel_dispatch" in t][0]
q = f"""select s.kernel_name, i.name, sum(p.value), count(distinct d.id) from {pmc} p join {info} i on p.pmc_id = i.id
        join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by 1, 2"""
acc = collections.defaultdict(dict)
for k, c, v, n in db.execute(q):
    acc[k][c] = v / n
for k, d in acc.items():
    if d.get("SQ_INSTS_LDS", 0) > 1000:
        print(f"{k[:50]:52s} conflict cycles {d.get('SQ_LDS_BANK_CONFLICT', 0):10.0f}  LDS instructions {d['SQ_INSTS_LDS']:9.0f}  per instruction {d.get('SQ_LDS_BANK_CONFLICT', 0) / d['SQ_INSTS_LDS']:.2f}  idx active {d.get('SQ_LDS_IDX_ACTIVE', 0):10.0f}")
PY
