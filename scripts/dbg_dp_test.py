"""Run tests/test_gpu_dp.py's one-rank engine test with a traceback dump if it stalls (python scripts/dbg_dp_test.py)."""
import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.dump_traceback_later(60, exit=True)
import test_gpu_dp
test_gpu_dp.test_owner_adam_on_one_rank_takes_the_plain_engines_steps()
print("ok")
