# the short-sweep rule of row-sharded data parallelism (optim.TableAdam.prefer_inline_for_short_sweeps): its tests, and the
# emulated per-rank step of an 8- / 4- / 1-rank shard with NOTHING pinned
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/r05_rule; mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_models.py tests/test_gpu_world2.py -m gpu -x -q -k "data_parallel or world2 or shard" > $OUT/pytest_dp.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_dp.log
for vs in 0.125 0.25 1.0; do
  timeout 200 python bench.py --force-dp --tables shard --vocab-scale $vs --steps 150 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting --no-pmc --no-twin-check > $OUT/auto_$vs.json 2> $OUT/auto_$vs.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/auto_$vs.json").read().strip().splitlines()[-1]); print("auto vocab-scale $vs", d["ms_per_step"], d["config"].get("step_form"), d.get("rows_behind_after_flush"))
except Exception as e:
    print("auto_$vs FAILED", e); print(open("$OUT/auto_$vs.err").read()[-800:])
PY
done
