#!/bin/bash
# Round-3 measurement session (one gpurun call): the driver's bench line, a 300-step line, rocprofv3 kernel stats of the
# driver's command, PMC traffic of the deferred window sweep (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
STAGES=${STAGES:-"bench bench300 prof pmc"}
for s in $STAGES; do case $s in
bench) echo "== bench default (driver's command)"; ( time timeout 600 python bench.py --steps 20 --warmup 10 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real; echo "rc=$?";;
bench300) echo "== bench 300"; timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief > "$OUT/bench_300.json" 2> "$OUT/bench_300.err"; echo "rc=$?";;
prof) echo "== rocprof kernel stats of the default command"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --brief > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
  python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1
  cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
  (cd /tmp && rm -rf /tmp/tl_r03 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_r03 -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> "$OLDPWD/$OUT/tl.err"); python tools/timeline.py /tmp/tl_r03 2 > "$OUT/step_timeline.txt" 2>&1
  find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete;;
pmc) for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sweep_$c" -o probe -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> "$OLDPWD/$OUT/pmc_sweep_$c.err"); echo "pmc $c rc=$?"
  python tools/prof_summary.py "$OUT/pmc_sweep_$c" --pmc $c --tail 25 > "$OUT/pmc_sweep_${c}.txt" 2>&1
  find "$OUT/pmc_sweep_$c" -name '*.csv' -size +5M -delete
 done;;
esac; done
echo "== done"
