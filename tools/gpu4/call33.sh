#!/bin/bash
# round 4, call 33: where experiment() spends its wall time at L outside the epochs (cProfile, after a warm-up call)
mkdir -p gpurun_out/r04
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r04/c33_experiment_profile.txt
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
import rgb_experiment_amd as R
from bench import MODELS, WORKLOADS, synth
wl = WORKLOADS["L"]
ei, x, y = synth(wl["N"], wl["E"], wl["d"])
data = R.Data(x=x, y=y, edge_index=ei)
params = dict(MODELS["gcn"][0])
kw = dict(specify_data=True, data=data, model_name="gcn", learning_rate=0.01, need_to_reappear=True, print_print=False)
R.experiment(params, epoch=1, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
R.experiment(params, epoch=20, **kw)
torch.cuda.synchronize(); pr.disable()
print("wall", time.perf_counter() - t0)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
PY
