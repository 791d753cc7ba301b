#!/bin/bash
# Every GPU test file in a process of its own under PYTORCH_NO_HIP_MEMORY_CACHING=1 (each tensor its own exact-size hipMalloc: an access past
# a buffer's end lands in unmapped memory instead of a neighbour's block).  Recording a hipGraph needs the caching allocator, so tests that
# capture FAIL here ("operation not permitted when stream is capturing") -- expected; what this run looks for is a memory access fault or an
# abort in the eager launches.  -> gpurun_out/r6_nocache_run.txt
out=gpurun_out/r6_nocache_run.txt
: > $out
for f in tests/test_gpu_*.py; do
  case $f in *test_gpu_bench.py|*test_gpu_sharded_two_ranks.py|*test_gpu_capture_guard.py) continue;; esac
  log=gpurun_out/nc_$(basename $f .py).log
  PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 400 python3 -m pytest $f -q -m gpu -p no:cacheprovider > $log 2>&1
  rc=$?
  nfault=$(grep -a -c "Memory access fault\|Aborted\|core dumped\|terminate called" $log)
  ncap=$(grep -a -c "when stream is capturing\|during capture\|graph capture" $log)
  echo "$f rc=$rc faults=$nfault capture_errors=$ncap :: $(grep -aE 'passed|failed' $log | tail -1)" >> $out
  if [ $rc -ge 124 ] || [ $nfault -gt 0 ]; then echo "STOP: $f rc=$rc" >> $out; break; fi
done
cat $out
