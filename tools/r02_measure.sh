#!/bin/bash
# Round-2 measurement session (one gpurun call): parity suite, the driver's bench line, the 300-step line, rocprofv3
# kernel stats of the driver's command, PMC traffic of the north-star kernels, the secondary configs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest"; timeout 600 python -m pytest tests/ -x -q -m gpu > "$OUT/pytest.log" 2>&1; grep -E "passed|failed" "$OUT/pytest.log" | tail -2
echo "== bench default (driver's command)"; timeout 400 python bench.py --steps 20 --warmup 10 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
echo "== bench 300"; timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline > "$OUT/bench_300.json" 2> "$OUT/bench_300.err"; echo "rc=$?"
echo "== rocprof kernel stats of the default command"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 10 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1
cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete
for B in ${PMC_BATCHES:-4096 65536}; do for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && PMC_B=$B timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_${B}_$c" -o probe -- python "$OLDPWD/tools/pmc_probe.py" > /dev/null 2> "$OLDPWD/$OUT/pmc_${B}_$c.err"); echo "pmc $B $c rc=$?"
  python tools/prof_summary.py "$OUT/pmc_${B}_$c" --pmc $c --tail 15 > "$OUT/pmc_${B}_${c}.txt" 2>&1
  find "$OUT/pmc_${B}_$c" -name '*.csv' -size +5M -delete
done; done
for m in dcnv2 din dssm; do echo "== bench $m"; timeout 300 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_$m.json" 2> "$OUT/bench_$m.err"; echo "rc=$?"; done
echo "== done"
