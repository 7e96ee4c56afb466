#!/bin/bash
# Usage: scripts/gaps_any.sh <tag> <script> [args...]   (run on the GPU box via gpurun)
# Kernel trace of bench.py -> GPU idle time between consecutive kernels, attributed to the kernel that follows the gap.
set -u
tag=${1:-gaps}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf /tmp/gaps_$tag "$out"; mkdir -p "$out"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_$tag -o $tag -- python3 "$@" > "$out/bench.log" 2>&1
f=$(find /tmp/gaps_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee "$out/gaps.txt"
import csv, sys, collections
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    if "at::native" in k:
        import re
        m = re.findall(r"(\w+Functor\w*|\w+_functor|direct_copy\w*|index_\w+|\w+_kernel\w*)", k)
        return "aten:" + (m[-1] if m else k[:40])
    return k.split("(")[0][:60]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps are delimited by sample_mask launches; keep the last 15 of them
marks = [i for i, r in enumerate(rows) if "sample_mask_kernel" in r[2]]
lo, hi = marks[-16], marks[-1]
span = rows[hi][0] - rows[lo][0]
busy = sum(e - s for s, e, _ in rows[lo:hi])
gap_by = collections.defaultdict(lambda: [0, 0]); dur_by = collections.defaultdict(lambda: [0, 0])
for i in range(lo + 1, hi + 1):
    g = rows[i][0] - max(r[1] for r in rows[max(lo, i - 4):i])
    k = rows[i][2]
    if g > 0: gap_by[k][0] += g; gap_by[k][1] += 1
for s, e, k in rows[lo:hi]:
    dur_by[k][0] += e - s; dur_by[k][1] += 1
n = 15
print("steps %d  span/step %.3f ms  busy/step %.3f ms  idle/step %.3f ms  launches/step %.1f" % (n, span / n / 1e6, busy / n / 1e6, (span - busy) / n / 1e6, (hi - lo) / n))
print("-- idle before kernel (per step)")
for k, (g, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:18]:
    print("%8.1f us  x%5.1f  %s" % (g / n / 1e3, c / n, k))
print("-- busy by kernel (per step)")
for k, (g, c) in sorted(dur_by.items(), key=lambda kv: -kv[1][0])[:30]:
    print("%8.1f us  x%5.1f  %s" % (g / n / 1e3, c / n, k))
PY
