#!/bin/bash
# profiles of the final library, then the full suite + bench line (what the driver runs at round end)
bash tools/gpu_profiles.sh > gpurun_out/profiles_run.log 2>&1
tail -2 gpurun_out/profiles_run.log
