#!/bin/bash
# Round driver for one gpurun call: smoke, GPU parity tests, bench (eager + hipGraph), rocprofv3 kernel stats.
# Every stage is bounded by its own timeout and logs under gpurun_out/ so a failure never hides the others.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$(date +%H%M%S)
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
STAGES=${STAGES:-"smoke kernels models bench prof"}
for s in $STAGES; do
  echo "=== stage $s $(date +%T)"
  case $s in
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "rc=$?" >> "$OUT/smoke.log"; tail -5 "$OUT/smoke.log";;
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 300 > "$OUT/kernels.log" 2>&1; echo "rc=$?" >> "$OUT/kernels.log"; tail -30 "$OUT/kernels.log";;
    kernels_all) timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 300 > "$OUT/kernels.log" 2>&1; echo "rc=$?" >> "$OUT/kernels.log"; tail -60 "$OUT/kernels.log";;
    models)  timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --timeout 300 > "$OUT/models.log" 2>&1; echo "rc=$?" >> "$OUT/models.log"; tail -40 "$OUT/models.log";;
    props)   timeout 900 python -m pytest tests/test_gpu_properties.py -q -m gpu --timeout 600 > "$OUT/props.log" 2>&1; echo "rc=$?" >> "$OUT/props.log"; tail -30 "$OUT/props.log";;
    bench)   timeout 900 python bench.py --steps ${BENCH_STEPS:-50} --warmup 10 --graph 0 > "$OUT/bench_eager.json" 2> "$OUT/bench_eager.err"; echo "rc=$?"; cat "$OUT/bench_eager.json"; tail -5 "$OUT/bench_eager.err"
             timeout 600 python bench.py --steps ${BENCH_STEPS:-50} --warmup 10 --graph 1 --no-cpu-baseline > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; echo "rc=$?"; cat "$OUT/bench_graph.json"; tail -5 "$OUT/bench_graph.err";;
    prof)    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps ${PROF_STEPS:-50} --warmup 10 --no-cpu-baseline ${PROF_BENCH_ARGS:-} > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
             python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1; head -45 "$OUT/prof_summary.txt"
             find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete;;
    pmc)     for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_$c" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --graph 0 ${PROF_BENCH_ARGS:-} > "$OLDPWD/$OUT/pmc_$c.json" 2> "$OLDPWD/$OUT/pmc_$c.err"); echo "pmc $c rc=$?"
               python tools/prof_summary.py "$OUT/pmc_$c" --pmc $c > "$OUT/pmc_${c}_summary.txt" 2>&1; head -30 "$OUT/pmc_${c}_summary.txt"
               find "$OUT/pmc_$c" -name '*.csv' -size +20M -delete
             done;;
    pmcprobe) for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/probe_$c" -o probe -- python "$OLDPWD/tools/pmc_probe.py" > /dev/null 2> "$OLDPWD/$OUT/probe_$c.err"); echo "probe $c rc=$?"
               python tools/prof_summary.py "$OUT/probe_$c" --pmc $c --tail 20 > "$OUT/probe_${c}_summary.txt" 2>&1; grep -E "rechub|adam_dense|kernel " "$OUT/probe_${c}_summary.txt"
               find "$OUT/probe_$c" -name '*.csv' -size +5M -delete
             done;;
    kbench)  timeout 600 python tools/kbench.py ${KBENCH_ARGS:-} > "$OUT/kbench.log" 2>&1; echo "rc=$?"; cat "$OUT/kbench.log";;
    *) echo "unknown stage $s";;
  esac
done
echo "=== done $(date +%T)"
