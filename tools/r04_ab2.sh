#!/bin/bash
# Same-box A/B with extra bench arguments: tools/r04_ab2.sh <outdir> "<tag>|<ENV=V;ENV=V...>|<bench args>" ...
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  IFS='|' read -r tag envs args <<< "$spec"
  envs=$(echo "$envs" | tr ';' ' ')
  env $envs timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting $args > $out/$tag.json 2> $out/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$out/$tag.json"))
    print("$tag", d["ms_per_step"], d["config"].get("step_form", {}).get("chosen"), d["config"].get("step_form", {}).get("ms_per_step_during_tuning"))
except Exception as e:
    print("$tag FAILED", e); print(open("$out/$tag.err").read()[-800:])
PY
done
