#!/bin/bash
# Usage (on the GPU box, via gpurun): bash run_gpu_checks.sh [tag]
# Runs the GPU parity tests, then the default bench; every GPU step has its own time limit.
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/gpu_tests_$tag.log
if [ $rc -ne 0 ]; then grep -E "FAILED|Error|error" gpurun_out/gpu_tests_$tag.log | head -20; exit $rc; fi
timeout -k 10 600 python -u bench.py "${@:2}" > gpurun_out/bench_$tag.log 2>&1
rc=$?
tail -2 gpurun_out/bench_$tag.log | cut -c1-1500
exit $rc
