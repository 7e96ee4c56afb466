#!/bin/bash
# A/B of the K-Planes backward schedules on one box: TN_BWD_SCHEDULE = fused | split | overlap (tinynerf_amd/fused.py)
mkdir -p gpurun_out/ab
for rep in 1 2; do
for s in fused split overlap; do
  TN_BWD_SCHEDULE=$s python scripts/step_config.py kplanes 60 2>&1 | tail -1 | sed "s/^/$s: /" | tee -a gpurun_out/ab/schedule.log
done
done
