# as tools/r05_session3.sh at 1/2 and 1/4 of the rows (2- and 4-rank shards): where the deferred sweep stops paying
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/r05_dp24; mkdir -p $OUT
run() { tag=$1; envs=$2; shift 2
  env $envs timeout 200 python bench.py --force-dp --tables shard --steps 150 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting --no-pmc --no-twin-check "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); print("$tag", d["ms_per_step"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-600:])
PY
}
for vs in 0.5 0.25; do
run s${vs}_deferred_k128 "X=1" --vocab-scale $vs
run s${vs}_deferred_k64 "X=1" --vocab-scale $vs --lazy-k 64
run s${vs}_inline_k128 "RECHUB_STEP_FORM=inline" --vocab-scale $vs
run s${vs}_inline_k64 "RECHUB_STEP_FORM=inline" --vocab-scale $vs --lazy-k 64
run s${vs}_inline_k32 "RECHUB_STEP_FORM=inline" --vocab-scale $vs --lazy-k 32
done
