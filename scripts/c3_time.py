"""config 3 alone (bench_legs.leg_config3): python scripts/c3_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
r = bench_legs.leg_config3()
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "ms_per_step_graph", "ms_per_step_eager") if k in r}))
