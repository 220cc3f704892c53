#!/bin/bash
# SQ counter passes of the bench workload at 0 % and 100 % on-target pairs
for ot in 0.0 1.0; do echo "== on-target $ot"; bash tools/gpu_pmc_one.sh $ot; done
