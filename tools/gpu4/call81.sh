#!/bin/bash
# round 4, call 81: the fuzz module and the ingest module as pytest runs them (pinned seeds), after the last edits
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_ingest.py -q -x --durations=5 2>&1 | tee gpurun_out/r04/c81_fuzz_ingest.log | tail -12 | cut -c1-300
exit 0
