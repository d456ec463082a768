#!/bin/bash
# Round-4 measurement session (one gpurun call; STAGES selects parts): GPU tests incl. the mounted-reference integration
# runs, the driver's bench line, a 300-step line, rocprofv3 kernel stats of the driver's command + one-step timeline,
# PMC traffic of the deferred window sweep (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only), per-model kernel
# tables, and the same-box A/B of the round's step changes.  Outputs under gpurun_out/r04/ (copied into profiles/ by hand).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
[ -d _refmount/torch_rechub ] && export RECHUB_REFERENCE=$PWD/_refmount
STAGES=${STAGES:-"tests bench bench300 prof pmc models ab hist"}
model_table() {  # kernel table of ONE traced step of a secondary config
  m=$1
  (cd /tmp && rm -rf /tmp/${m}_prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${m}_prof -o $m -- python $OLDPWD/bench.py --model $m --trace-inner --steps 10 --warmup 5 > /dev/null 2> $OLDPWD/$OUT/${m}_prof.err)
  python - $m <<'PY'
import csv,glob,os,sys
m=sys.argv[1]
f=max(glob.glob(f'/tmp/{m}_prof/**/*kernel_trace.csv',recursive=True),key=os.path.getsize)
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks=[i for i,r in enumerate(rows) if "batch_gather_kernel" in r[2] or "refresh_assemble" in r[2]]
if sum(1 for r in rows if "step_ahead" in r[2])>len(marks): marks=[i for i,r in enumerate(rows) if "embed_fwd_kernel" in r[2]]
lo,hi=marks[-3],marks[-2]
agg={}
for st,en,n in rows[lo:hi]:
    k=n.replace("void ","").replace("(anonymous namespace)::","").replace("rechub::","").split("(")[0][:70]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(en-st)/1e3
print(m, "step wall us", (rows[hi][0]-rows[lo][0])/1e3, "busy", round(sum(v[1] for v in agg.values()),1), "launches", sum(v[0] for v in agg.values()))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:30]: print("%-72s %3d %9.1f"%(k,v[0],v[1]))
PY
}
for s in $STAGES; do case $s in
tests) echo "== pytest -m gpu (reference mounted: ${RECHUB_REFERENCE:-no})"
  timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log"
  timeout 300 python -m pytest tests/test_integration_patch.py -m gpu -v > "$OUT/integration_gpu.log" 2>&1; grep -E "PASSED|FAILED|SKIPPED|passed|failed" "$OUT/integration_gpu.log" | tail -12;;
bench) echo "== bench default (driver's command)"; ( time timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real; echo "rc=$?";;
bench300) echo "== bench 300"; timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief > "$OUT/bench_300.json" 2> "$OUT/bench_300.err"; echo "rc=$?";;
prof) echo "== rocprof kernel stats of the default command"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --brief > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
  python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1
  cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
  (cd /tmp && rm -rf /tmp/tl_r04 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_r04 -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> "$OLDPWD/$OUT/tl.err"); python tools/timeline.py /tmp/tl_r04 2 > "$OUT/step_timeline.txt" 2>&1
  find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete;;
pmc) for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sweep_$c" -o probe -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> "$OLDPWD/$OUT/pmc_sweep_$c.err"); echo "pmc $c rc=$?"
  python tools/prof_summary.py "$OUT/pmc_sweep_$c" --pmc $c --tail 25 > "$OUT/pmc_sweep_${c}.txt" 2>&1
  find "$OUT/pmc_sweep_$c" -name '*.csv' -size +5M -delete
 done;;
models) for m in dcnv2 din dssm; do model_table $m > "$OUT/${m}_step_kernels.txt" 2>&1; head -3 "$OUT/${m}_step_kernels.txt"; done;;
ab) echo "== same-box A/B (200 steps each, two rounds)"
  for r in 1 2; do bash tools/r04_ab2.sh $OUT/ab "all_on_$r||" "eager_head_$r|RECHUB_AB=ahead=0;RECHUB_TUNE=12=6000|" "strict_join_$r|RECHUB_AB=lookahead=0|" "chain_off_$r|RECHUB_AB=chain=0|" "wgrad_206reg_$r|RECHUB_TUNE=11=0|" "lazy_k64_$r||--lazy-k 64" "round3_$r|RECHUB_AB=lookahead=0,assemble=0,chain=0,headside=0;RECHUB_TUNE=11=0|--lazy-k 64"; done 2>&1 | tee "$OUT/ab.txt";;
hist) echo "== step period distribution without a profiler (tools/period_hist.py, 400 steps)"
  bash tools/r04_period.sh ahead= 'eager_head=RECHUB_AB=ahead=0;RECHUB_TUNE=12=6000' strict_join=RECHUB_AB=lookahead=0 2>&1 | tee "$OUT/period_hist.txt";;
esac; done
echo "== done"
