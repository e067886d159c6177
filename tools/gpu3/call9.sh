#!/bin/bash
# round 3, call 9: launches per epoch at S after the fold / fused Adam / merged scaling; default bench line at L
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cross_entropy or frozen or hip_graph or model_logits or experiment" > $O/c9_tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/c9_tests.log)"
for S in 10 20; do
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/c9_S_s$S -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload S --no-cpu-baseline --primary-only --steps $S --warmup 3 > $GRAFT_REPO_ROOT/$O/c9_S_s$S.json 2> $GRAFT_REPO_ROOT/$O/c9_S_s$S.log)
done
python tools/epoch_diff.py $O/c9_S_s10 10 $O/c9_S_s20 20 --out $O/c9_S_epoch.csv --top 60 | head -2
timeout -k 10 600 python bench.py > $O/c9_bench.json 2> $O/c9_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/c9_bench.json') if l.startswith('{')][-1])
print('L', round(d['ms_per_step'],2), 'median', round(d['median_ms_per_step'],2), 'frac', round(d['roofline']['frac'],3), 'parity', d['parity']['sampled_logits']['max_abs_diff_hip_vs_oracle'], d['cpu_baseline']['parity_at_full_size']['max_abs_diff_hip_vs_cpu'])
for k in ('configs_1_same_run','configs_0_same_run','undirected_same_run','powerlaw_same_run','train_step_only','hip_graph_replay'):
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in d[k].items() if a in ('ms_per_step','median_ms_per_step','hip_graph_replay','eager_ms_per_epoch','hip_graph_ms_per_epoch','roofline_frac_algorithmic','error')})
PY
