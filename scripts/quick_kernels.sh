#!/bin/bash
# per-kernel durations of the headline step alone (rocprofv3 kernel trace of bench.py without its other legs)
bash scripts/prof_any.sh ${1:-qk} $PWD/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-c5 --no-legs --no-extras --no-baselines | grep -E "enc_tile_step|enc_tail_k|enc_grad_reduce|enc_step_k|step_stage|batch_prep"
