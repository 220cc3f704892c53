#!/bin/bash
# profiles of the final library, then the landscape / panel tables of profiles/README.md
bash tools/gpu_profiles.sh > gpurun_out/profiles_run.log 2>&1
tail -2 gpurun_out/profiles_run.log
mkdir -p gpurun_out/final
bash tools/gpu_landscape.sh 2>&1 | tee gpurun_out/final/landscape.txt
bash tools/gpu_panel.sh "10 30 60 100 150" base 2>&1 | tee gpurun_out/final/panel.txt
bash tools/gpu_ab2.sh base 2>&1 | tee gpurun_out/final/ontarget.txt
python tests/scale_check.py --k 31 --q 20 --single --bf-log2 37 --skip-bitvector --oracle-pairs 0 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('configs4 shape', d['mode'], d['kernel_ms'], d['n_assoc'])" | tee -a gpurun_out/final/landscape.txt
for ot in 0.0 1.0; do python tests/scale_check.py --skip-bitvector --oracle-pairs 0 --on-target $ot 2>/dev/null | grep kernel_ms | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('configs2 index ot=$ot', d['mode'], d['kernel_ms'], d['n_assoc'])" | tee -a gpurun_out/final/landscape.txt; done
