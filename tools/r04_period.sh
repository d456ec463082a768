#!/bin/bash
# Step period distribution WITHOUT a profiler (tools/period_hist.py) per configuration: r04_look4.sh "<tag>=<ENV=V;ENV=V>" ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for spec in "$@"; do
  tag=${spec%%=*}; envs=${spec#*=}; [ "$envs" = "$spec" ] && envs=""
  envs=$(echo "$envs" | tr ';' ' ')
  echo "== $tag: $(env $envs timeout 300 python tools/period_hist.py --steps 400 --warmup 10 2>&1 | tail -1)"
done
