#!/bin/bash
# the exchange-wait events and the N > 1 line fields on the rehearsal path (ranks sharing the GPU over gloo), + dist tests
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
export RGBX_DIST_BACKEND=gloo
timeout -k 10 300 python bench.py --gpus 4 --workload S --steps 3 --warmup 1 --no-cpu-baseline > $O/rehearse25_4.json 2> $O/rehearse25_4.err
echo "gpus=4 rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rehearse25_4.json') if l.startswith('{')][-1]); print(d['scheme'], d['kernel_ms_by_kind'].get('exchange_wait'), [r['exposed_exchange_ms_per_step'] for r in d['per_rank']], d['small_all_to_all_us_measured'], d['link_gbs_measured'])" 2>&1 | tail -1)"
timeout -k 10 300 python bench.py --gpus 2 --workload S --steps 3 --warmup 1 --no-cpu-baseline --exchange 1x2 > $O/rehearse25_2.json 2> $O/rehearse25_2.err
echo "gpus=2 rc=$? $(python -c "import json; d=json.loads([l for l in open('$O/rehearse25_2.json') if l.startswith('{')][-1]); print(d['scheme'], d['kernel_ms_by_kind'].get('exchange_wait'), [r['exposed_exchange_ms_per_step'] for r in d['per_rank']])" 2>&1 | tail -1)"
