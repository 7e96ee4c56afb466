#!/bin/bash
# Usage: scripts/profile_config.sh <method> [steps]   (GPU box) -- kernel-trace stats + matrix-pipe counters of one model
# configuration's training step (scripts/step_config.py); summaries land in gpurun_out/cfg_<method>/.
set -u
m=$1; steps=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cfg_$m; rm -rf "$out" /tmp/pc_$m /tmp/pcm_$m; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$m -o $m -- python3 scripts/step_config.py $m $steps > "$out/run.log" 2>&1
find /tmp/pc_$m -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
bash scripts/pmc_any.sh cfgm_$m "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" scripts/step_config.py $m 4 > "$out/mfma.log" 2>&1
cp gpurun_out/cfgm_$m/pmc.json "$out/mfma_busy.json"
tail -1 "$out/run.log"; head -8 "$out/kernel_stats.csv" | cut -c1-150
