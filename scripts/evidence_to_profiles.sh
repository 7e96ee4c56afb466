#!/bin/bash
# Usage (here, after `gpurun -- bash scripts/evidence.sh <tag>`): scripts/evidence_to_profiles.sh <tag> <round prefix, e.g. round3>
set -eu
tag=$1; pre=$2
cp gpurun_out/$tag/bench.json profiles/${pre}_bench.json
cp gpurun_out/${tag}_prof/${tag}_prof_kernel_stats.csv profiles/${pre}_step_kernel_stats.csv
samples=$(python3 -c "import json;print(json.loads(open('gpurun_out/$tag/bench.json').read().strip().splitlines()[-1])['config']['samples_per_step_per_gpu'])")
python3 scripts/pmc_to_json.py gpurun_out/${tag}_pmc profiles/${pre}_pmc_traffic.json $samples
cp gpurun_out/${tag}_mfma/mfma.json profiles/${pre}_mfma_busy.json
for m in vanilla cobafa; do
  cp gpurun_out/cfg_$m/kernel_stats.csv profiles/${pre}_${m}_kernel_stats.csv
  cp gpurun_out/cfg_$m/mfma_busy.json profiles/${pre}_${m}_mfma_busy.json
done
python3 - <<PY
import json
d=json.loads(open('profiles/${pre}_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['other_configs'].items()})
t=json.load(open('profiles/${pre}_pmc_traffic.json'))
print('traffic keys', list(t.keys())[:6])
PY
