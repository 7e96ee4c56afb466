#!/bin/bash
# Usage (here, after `gpurun -- bash scripts/evidence.sh <tag>`): scripts/evidence_to_profiles.sh <tag> <round prefix, e.g. round3>
set -eu
tag=$1; pre=$2
cp gpurun_out/$tag/bench.json profiles/${pre}_bench.json
cp gpurun_out/${tag}_prof/${tag}_prof_kernel_stats.csv profiles/${pre}_step_kernel_stats.csv
samples=$(python3 -c "import json;print(json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1])['config']['samples_per_step_per_gpu'])")
python3 scripts/pmc_to_json.py gpurun_out/${tag}_pmc profiles/${pre}_pmc_traffic.json $samples
head=$(git log -1 --format=%h)
wrap() {   # raw per-kernel counter averages -> {"note": ..., "per_kernel": {...}} (the form bench.py reads)
python3 - "$1" "$2" "$3" <<'PY'
import json, sys
src, dst, note = sys.argv[1:4]
d = json.load(open(src))
d = d.get("per_kernel", d)
json.dump({"note": note + ": rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace, averages per "
           "launch.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (v_mfma_f32_32x32x2_f32 = 64 busy cycles, "
           "v_mfma_f32_32x32x16_bf16 = 32), GRBM_GUI_ACTIVE over the 8 XCDs: mfma_pipe_busy_frac = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).",
           "per_kernel": d}, open(dst, "w"), indent=1)
PY
}
wrap gpurun_out/${tag}_mfma/mfma.json profiles/${pre}_mfma_busy.json "scripts/pmc_mfma.sh over bench.py --no-cpu-baseline --no-stages --steps 20 at $head"
for m in vanilla cobafa; do
  cp gpurun_out/cfg_$m/kernel_stats.csv profiles/${pre}_${m}_kernel_stats.csv
  wrap gpurun_out/cfg_$m/mfma_busy.json profiles/${pre}_${m}_mfma_busy.json "scripts/profile_config.sh $m (scripts/step_config.py $m 4) at $head"
done
for m in vanilla cobafa; do cp gpurun_out/pmc_$m/traffic.json profiles/${pre}_${m}_pmc_traffic.json; done
cp gpurun_out/$tag/full_recipe.json profiles/${pre}_full_recipe.json
python3 - <<PY
import json
d=json.loads(open('profiles/${pre}_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['other_configs'].items()})
t=json.load(open('profiles/${pre}_pmc_traffic.json'))
print('traffic keys', list(t.keys())[:6])
PY
