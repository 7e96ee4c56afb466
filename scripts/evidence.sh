#!/bin/bash
# Usage (GPU box, via gpurun): scripts/evidence.sh <tag>  -- everything profiles/<round>_* is made from, at the current commit:
# the GPU test suite, the default bench line, its kernel-trace stats, the HBM-traffic and matrix-pipe counter passes (each in
# its own rocprofv3 run, kernel-trace only), and the per-kernel stats of the Vanilla / Cobafa side configurations.
set -u
tag=${1:-ev}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1; tail -2 gpurun_out/$tag/tests.log
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; tail -c 300 gpurun_out/$tag/bench.json; echo
bash scripts/profile.sh ${tag}_prof > gpurun_out/$tag/profile.log 2>&1; grep "total kernel" gpurun_out/$tag/profile.log
bash scripts/pmc.sh ${tag}_pmc --no-stages --steps 20 > gpurun_out/$tag/pmc.log 2>&1; tail -3 gpurun_out/$tag/pmc.log
bash scripts/pmc_mfma.sh ${tag}_mfma --no-stages --steps 20 > gpurun_out/$tag/mfma.log 2>&1; tail -3 gpurun_out/$tag/mfma.log
bash scripts/profile_config.sh vanilla 8 > gpurun_out/$tag/cfg_vanilla.log 2>&1; tail -3 gpurun_out/$tag/cfg_vanilla.log | cut -c1-160
bash scripts/profile_config.sh cobafa 8 > gpurun_out/$tag/cfg_cobafa.log 2>&1; tail -3 gpurun_out/$tag/cfg_cobafa.log | cut -c1-160
bash scripts/pmc_config.sh vanilla 6 > gpurun_out/$tag/pmc_vanilla.log 2>&1; tail -14 gpurun_out/$tag/pmc_vanilla.log | cut -c1-160
bash scripts/pmc_config.sh cobafa 6 > gpurun_out/$tag/pmc_cobafa.log 2>&1; tail -14 gpurun_out/$tag/pmc_cobafa.log | cut -c1-160
python bench.py --full-recipe > gpurun_out/$tag/full_recipe.json 2> gpurun_out/$tag/full_recipe.err; tail -c 600 gpurun_out/$tag/full_recipe.json; echo
