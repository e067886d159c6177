#!/bin/bash
# round 3, call 1: new kernel modes (dense / blocked), frozen-weight loss epilogue, APPNP K=10 at S and L, then the
# whole -m gpu suite and the default bench line
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dense_mode or blocked_exchange or frozen" > $O/c1_new.log 2>&1
echo "new tests rc=$? $(tail -1 $O/c1_new.log)"
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "appnp_k10" > $O/c1_k10.log 2>&1
echo "k10 rc=$? $(tail -1 $O/c1_k10.log)"
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py::test_appnp_k10_on_the_whole_benchmark_graph > $O/c1_all.log 2>&1
echo "all rc=$? $(tail -1 $O/c1_all.log)"
timeout -k 10 600 python bench.py > $O/c1_bench.json 2> $O/c1_bench.err
echo "bench rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/c1_bench.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), d['median_ms_per_step'], d['roofline']['frac'])")"
