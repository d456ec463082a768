# replicated tables: per-rank compute of a 4- / 8-rank job emulated on one rank (RECHUB_EMULATE_WORLD=N: the gathers return N
# copies of the local rows, so the scatter and the touched pass see the row volume of N ranks; timing only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/r05_rep; mkdir -p $OUT
for n in 8 4; do
  RECHUB_EMULATE_WORLD=$n timeout 150 python bench.py --force-dp --tables replicate --steps 150 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting --no-pmc --no-twin-check > $OUT/rep_$n.json 2> $OUT/rep_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/rep_$n.json").read().strip().splitlines()[-1]); print("replicate emulate_world $n", d["ms_per_step"], d["config"].get("step_form"))
except Exception as e:
    print("rep_$n FAILED", e); print(open("$OUT/rep_$n.err").read()[-600:])
PY
done
