#!/bin/bash
# round 4, call 8: contended emulation of rank 0 of 8 (gcn) + the single-GPU line of the same box
set -o pipefail
mkdir -p gpurun_out/r04
O=gpurun_out/r04
python bench.py --emulate-rank 8 --steps 12 --warmup 3 --no-cpu-baseline > $O/c8_emu8_gcn.json 2> $O/c8_emu8_gcn.err; echo "rc=$?"
tail -2 $O/c8_emu8_gcn.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/c8_emu8_gcn.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "scheme", d.get("scheme"))
e=d["emulated"]
print(json.dumps(e.get("contended"), indent=1)[:3000])
print(json.dumps(e["schedule_replay"]["by_link_rate"] if "by_link_rate" in e["schedule_replay"] else e["schedule_replay"], indent=0)[:1500])
PY
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --primary-only > $O/c8_single.json 2>> $O/c8_emu8_gcn.err; python -c "
import json;d=json.loads(open('gpurun_out/r04/c8_single.json').read().strip().splitlines()[-1]);print('single-GPU ms_per_step',d['ms_per_step'],d['kernel_ms_by_variant'])"
