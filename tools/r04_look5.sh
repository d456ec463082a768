#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04look
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "full_size_graph_step or graph_mode_flush" > "$OUT/pytest_look.log" 2>&1; echo "rc=$?"; tail -3 "$OUT/pytest_look.log"
for tag in ${TAGS:-ahead}; do
  (cd /tmp && rm -rf /tmp/tl_$tag && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 160 --warmup 10 > /dev/null 2> "$OLDPWD/$OUT/tl_$tag.err")
  echo "== $tag"; python tools/step_stats.py /tmp/tl_$tag 150 2>&1 | tee "$OUT/step_stats_$tag.txt"
  python tools/timeline.py /tmp/tl_$tag 1 | head -24
done
bash tools/r04_look4.sh g16=RECHUB_TUNE=13=16000 g22=RECHUB_TUNE=13=22000 g28= g34=RECHUB_TUNE=13=34000 g40=RECHUB_TUNE=13=40000
for r in 1 2; do bash tools/r04_ab.sh $OUT/ab4 ahead_$r= relaxed_$r=RECHUB_AB=ahead=0,RECHUB_TUNE=12=6000; done
