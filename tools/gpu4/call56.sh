#!/bin/bash
# round 4, call 56: the emulated rank of 8 on the code as it stands (cached operands wait across streams now)
mkdir -p gpurun_out/r04
timeout -k 10 600 python bench.py --emulate-rank 8 --no-cpu-baseline > gpurun_out/r04/c56_emulated_P8.json 2> gpurun_out/r04/c56_emulated_P8.err
echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/c56_emulated_P8.json").read().strip().splitlines()[-1])
e = d["emulated"]
print("compute", d["ms_per_step"], d["median_ms_per_step"], "host enqueue", d["per_rank"][0].get("host_enqueue_ms_per_step"))
for k in ("contended", "contended_cache_bypassing_traffic", "contended_control_sync_only"):
    c = e.get(k)
    if c: print(k, {kk: c[kk] for kk in c if "ms" in kk and not isinstance(c[kk], (list, dict))})
print("replay", json.dumps(e.get("schedule_replay", {}).get("by_link_rate"))[:300])
PY
tail -2 gpurun_out/r04/c56_emulated_P8.err | cut -c1-200
exit 0
