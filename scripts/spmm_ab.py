"""SpMM launch time with another build of the library: python scripts/spmm_ab.py librecengine_x.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recboard_amd import lib
if len(sys.argv) > 1:
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", sys.argv[1])
sys.argv = sys.argv[:1]
exec(open(os.path.join(ROOT, "scripts", "spmm_time.py")).read())
