"""kernel durations and the gaps in front of them over the last dispatches of a rocprofv3 kernel trace (gpurun_out/<tag>/*.db)"""
import glob
import sqlite3
import sys

import numpy as np
db = sqlite3.connect(glob.glob(f"gpurun_out/{sys.argv[1]}/*.db")[0])
rows = list(db.execute("select name,start,end from kernels order by start"))
names = [r[0][:34] for r in rows]; st = np.array([r[1] for r in rows]); en = np.array([r[2] for r in rows])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
i1 = len(rows) - 1
while i1 > 0 and "enc_grad_reduce" not in names[i1]:
    i1 -= 1
for i in range(max(1, i1 - n), i1 + 1):
    print(f"{names[i]:36s} dur {(en[i]-st[i])/1e3:6.1f}  gap_before {(st[i]-en[i-1])/1e3:6.1f}")
