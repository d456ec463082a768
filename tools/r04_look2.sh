#!/bin/bash
# Round-4 relaxed-join session: bit-equality tests of the step forms, per-step statistics (tools/step_stats.py) of the relaxed
# form at several hold-backs and of the strict form, same-box A/B (200 steps each, two rounds).  Outputs: gpurun_out/r04look/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04look
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -q -x -k "full_size_graph_step or graph_mode_flush or chain" > "$OUT/pytest_look.log" 2>&1; echo "rc=$?"; tail -3 "$OUT/pytest_look.log"
for tag in ${TAGS:-ahead ahead_3us ahead_10us relaxed strict}; do
  unset RECHUB_AB RECHUB_TUNE
  case $tag in strict) export RECHUB_AB=lookahead=0;; relaxed) export RECHUB_AB=ahead=0;; ahead_3us) export RECHUB_TUNE=12=3000;; ahead_10us) export RECHUB_TUNE=12=10000;; esac
  (cd /tmp && rm -rf /tmp/tl_$tag && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 160 --warmup 10 > /dev/null 2> "$OLDPWD/$OUT/tl_$tag.err")
  echo "== $tag"; python tools/step_stats.py /tmp/tl_$tag 150 2>&1 | tee "$OUT/step_stats_$tag.txt" | grep -v "^bn_\|^linear\|^embed_bwd\|^pack\|false, false, false"
done
unset RECHUB_AB RECHUB_TUNE
for r in 1 2; do bash tools/r04_ab.sh $OUT/ab ahead_$r= relaxed_$r=RECHUB_AB=ahead=0 strict_$r=RECHUB_AB=lookahead=0; done 2>&1 | tee "$OUT/ab.txt"
