#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04look
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for tag in ${TAGS:-ahead}; do
  (cd /tmp && rm -rf /tmp/tl_$tag && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 160 --warmup 10 > /dev/null 2> "$OLDPWD/$OUT/tl_$tag.err")
  echo "== $tag"; python tools/step_stats.py /tmp/tl_$tag 150 2>&1 | tee "$OUT/step_stats_$tag.txt"
  python tools/timeline.py /tmp/tl_$tag 2 | head -40
done
bash tools/r04_look4.sh ahead= relaxed=RECHUB_AB=ahead=0
for r in 1 2; do bash tools/r04_ab.sh $OUT/ab4 ahead_$r= relaxed_$r=RECHUB_AB=ahead=0; done
