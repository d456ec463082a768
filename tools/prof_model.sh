#!/bin/bash
# rocprofv3 kernel stats of one secondary config:  bash tools/prof_model.sh din [steps]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
M=${1:-din}; K=${2:-10}
OUT=gpurun_out/prof_$M
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --model $M --steps $K --warmup 2 --no-cpu-baseline > "$OLDPWD/$OUT/bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
python tools/prof_summary.py "$OUT/prof" > "$OUT/summary.txt" 2>&1
find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete
head -60 "$OUT/summary.txt"
