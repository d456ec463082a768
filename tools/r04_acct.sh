#!/bin/bash
# In-graph kernel table of the headline step (bench.py's nested rocprofv3 pass): tools/r04_acct.sh <outdir> <tag>=<ENV=V,..> ...
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  tag=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
  envs=$(echo "$envs" | tr ',' ' ')
  env $envs timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-sweep --acct-only > $out/$tag.json 2> $out/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$out/$tag.json"))
    a = d.get("step_accounting") or {}
    print("== $tag", d["ms_per_step"], "wall_us", a.get("wall_us_per_step"), "launches", a.get("kernel_launches_per_step"))
    for k in a.get("kernels", []):
        print("   %-70s x%.1f %7.2f us" % (k["kernel"][:70], k["launches_per_step"], k["us_per_step"]))
    print("   groups", a.get("groups_us_per_step"))
except Exception as e:
    print("$tag FAILED", e); print(open("$out/$tag.err").read()[-1500:])
PY
done
