#!/bin/bash
# round 4, call 65: the GAT row-split scratch with both F-sized arrays first (gat.hip): the new hub test, every gat test, the
# conv fuzz, the model fuzz seed that found it
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "gat or odd_number_of_hub or conv_layers_against" 2>&1 | tail -5 | cut -c1-300
timeout -k 10 300 python tools/fuzz_soak.py --models 280 290 2>&1 | grep -v "amdgpu.ids" | tail -4 | cut -c1-400
exit 0
