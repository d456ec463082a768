# per-rank compute of an 8-rank row-sharded job, emulated on one rank: every table at 1/8 of its rows (--vocab-scale 0.125: the
# shard's sweep, Adam state and window), the full per-rank batch; step form x lazy_k (the data-parallel step does not self-tune)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/r05_dp8; mkdir -p $OUT
run() { tag=$1; envs=$2; shift 2
  env $envs timeout 200 python bench.py --force-dp --tables shard --vocab-scale 0.125 --steps 150 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting --no-pmc --no-twin-check "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); print("$tag", d["ms_per_step"], d["config"].get("step_form"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/$tag.err").read()[-600:])
PY
}
run deferred_k128 "X=1"
run inline_k128 "RECHUB_STEP_FORM=inline"
run inline_k32 "RECHUB_STEP_FORM=inline" --lazy-k 32
run deferred_k32 "X=1" --lazy-k 32
run inline_k64 "RECHUB_STEP_FORM=inline" --lazy-k 64
run inline_k16 "RECHUB_STEP_FORM=inline" --lazy-k 16
