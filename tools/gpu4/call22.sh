#!/bin/bash
# round 4, call 22: BASELINE config 3 (GAT, 8 heads, workload S): bench line + per-kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp
python bench.py --workload S --model gat --steps 30 --warmup 5 > $O/rec_bench_S_gat.json 2> $O/rec_bench_S_gat.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/rec_bench_S_gat.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['median_ms_per_step'], d.get('hip_graph_replay'), d.get('train_step_only'))
print({k:(v['n'],round(v['avg_ms'],4)) for k,v in d['kernel_ms_by_kind'].items()})
PY
rm -rf $O/stats_S_gat
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats_S_gat -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload S --model gat --primary-only --no-cpu-baseline --steps 30 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$O/stats_S_gat.log)
rm -f $O/stats_S_gat/p_kernel_trace.csv
python - <<'PY'
import csv,glob,re
f=glob.glob('gpurun_out/r04/stats_S_gat/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n=re.sub(r'\(.*','',r['Name']); n=re.sub(r'void |rgbx::\(anonymous namespace\)::','',n)[:70]
    print(f"{n:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f} %")
print("total kernel ms per step (35 steps incl. warm-up):", tot/1e6/35)
PY
