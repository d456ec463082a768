#!/bin/bash
# Round-5 measurement session (one gpurun call; STAGES selects parts).  Outputs under gpurun_out/r05/ (the summaries that are
# cited are copied into profiles/ by hand).
#   tests      the driver's command: pytest tests -m gpu -x -q            (TESTS_REPEAT=n runs it n times)
#   testsall   the same without -x (every failure listed)
#   probe      tools/bitwise_probe.py: which HIP-vs-HIP pairs are bit-exact on order-free data
#   noise      tools/noise_budget.py: run-to-run spread of one training under float atomics (outlier budget of the suite)
#   bench      the driver's bench command (incl. CPU baseline, dense twin check)
#   bench300   300 steps, no CPU baseline
#   dp         one-rank data-parallel step (RCCL on a world of one), both placements
#   prof       rocprofv3 kernel stats of the default command + one-step timeline
#   pmc        FETCH_SIZE / WRITE_SIZE passes (kernel trace only) of the steady-state step
#   models     per-kernel tables of one traced step of dcnv2 / din / dssm
#   ab         same-box A/B lines: AB_CASES="name|ENV=..;ENV2=..|bench args" ...
#   hist       step period distribution without a profiler (tools/period_hist.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05${TAG:+_$TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
STAGES=${STAGES:-"tests bench"}
model_table() {
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
for s in $STAGES; do echo "=== stage $s $(date +%T)"; case $s in
tests) for i in $(seq 1 ${TESTS_REPEAT:-1}); do
    timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu_$i.log" 2>&1; echo "run $i rc=$?"; tail -3 "$OUT/pytest_gpu_$i.log"; done;;
testsall) timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu_all.log" 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu_all.log" | tail -40;;
probe) timeout 600 python tools/bitwise_probe.py ${PROBE_ARGS:-} > "$OUT/bitwise_probe.txt" 2> "$OUT/bitwise_probe.err"; echo "rc=$?"; grep -v "^ " "$OUT/bitwise_probe.txt" | tail -40; tail -5 "$OUT/bitwise_probe.err";;
noise) timeout 600 python tools/noise_budget.py --runs ${NOISE_RUNS:-24} > "$OUT/noise_budget.txt" 2> "$OUT/noise_budget.err"; echo "rc=$?"; tail -30 "$OUT/noise_budget.txt"; tail -3 "$OUT/noise_budget.err";;
bench) ( time timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real; echo "rc=$?"
  python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","flush_ms","rows_behind_after_flush","dense_twin_check")})
print("roofline", {k:d["roofline"].get(k) for k in ("bound","kernel","achieved","peak","frac","avg_launch_ms","traffic")})
print("sec", {k:(v.get("ms_per_step") if isinstance(v,dict) else v) for k,v in (d.get("secondary_configs") or {}).items()})
acct=d.get("step_accounting") or {}
print("acct", {k:acct.get(k) for k in ("wall_us_per_step","groups")})
PY
  tail -5 "$OUT/bench_default.err";;
bench300) timeout 400 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief ${BENCH300_ARGS:-} > "$OUT/bench_300.json" 2> "$OUT/bench_300.err"; echo "rc=$?"; python -c "
import json,sys; d=json.loads(open('$OUT/bench_300.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('dense_twin_check'))";;
dp) for pl in replicate shard; do
    timeout 400 python bench.py --force-dp --tables $pl --steps 100 --warmup 10 --no-cpu-baseline --brief > "$OUT/bench_dp_$pl.json" 2> "$OUT/bench_dp_$pl.err"; echo "dp $pl rc=$?"
    python -c "
import json,sys; d=json.loads(open('$OUT/bench_dp_$pl.json').read().strip().splitlines()[-1]); print('$pl', d['value'], d['ms_per_step'], d.get('comm_us_per_step'))" || tail -5 "$OUT/bench_dp_$pl.err"; done;;
prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --brief --no-twin-check > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
  python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1; head -30 "$OUT/prof_summary.txt"
  cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
  (cd /tmp && rm -rf /tmp/tl_r05 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_r05 -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> "$OLDPWD/$OUT/tl.err"); python tools/timeline.py /tmp/tl_r05 2 > "$OUT/step_timeline.txt" 2>&1
  find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete;;
pmc) for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && time timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc_sweep_$c" -o probe -- python "$OLDPWD/bench.py" --pmc-inner > /dev/null 2> "$OLDPWD/$OUT/pmc_sweep_$c.err"); echo "pmc $c rc=$?"
  python tools/prof_summary.py "$OUT/pmc_sweep_$c" --pmc $c --tail 25 > "$OUT/pmc_sweep_${c}.txt" 2>&1
  find "$OUT/pmc_sweep_$c" -name '*.csv' -size +5M -delete
 done;;
dptrace) for pl in ${DP_PLACEMENTS:-shard replicate}; do
  (cd /tmp && rm -rf /tmp/dp_prof_$pl && MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dp_prof_$pl -o dp -- python $OLDPWD/bench.py --force-dp --tables $pl --trace-inner --steps 10 --warmup 5 --rows 4000000 ${DP_ARGS:-} > /dev/null 2> $OLDPWD/$OUT/dp_prof_$pl.err)
  python - $pl > "$OUT/dp_${pl}_step_kernels.txt" 2>&1 <<'PY'
import csv,glob,os,sys
m=sys.argv[1]
f=max(glob.glob(f'/tmp/dp_prof_{m}/**/*kernel_trace.csv',recursive=True),key=os.path.getsize)
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","")) for r in csv.DictReader(open(f))]
rows.sort()
marks=[i for i,r in enumerate(rows) if "batch_gather_kernel" in r[2] or "refresh_assemble" in r[2]]
lo,hi=marks[-3],marks[-2]
print(m, "step wall us", (rows[hi][0]-rows[lo][0])/1e3, "launches", hi-lo)
t0=rows[lo][0]
for st,en,n,q in rows[lo:hi]:
    k=n.replace("void ","").replace("(anonymous namespace)::","").replace("rechub::","").split("(")[0][:64]
    print("%9.1f %9.1f %7.1f q%s %s"%((st-t0)/1e3,(en-t0)/1e3,(en-st)/1e3,q,k))
PY
  head -60 "$OUT/dp_${pl}_step_kernels.txt"; done;;
sweepalone) timeout 300 python tools/sweep_alone_probe.py ${SWEEP_K:-128} > "$OUT/sweep_alone.txt" 2> "$OUT/sweep_alone.err"; echo "rc=$?"; cat "$OUT/sweep_alone.txt";;
twin) timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --brief --twin-repeat --no-kernel-sweep > "$OUT/bench_twin.json" 2> "$OUT/bench_twin.err"; echo "rc=$?"; grep "dense twin check" "$OUT/bench_twin.err" | tail -2; tail -3 "$OUT/bench_twin.err";;
sec) for m in ${MODELS:-dssm dcnv2 din}; do
    timeout 400 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --brief > "$OUT/bench_$m.json" 2> "$OUT/bench_$m.err"; echo "$m rc=$?"
    python -c "
import json; d=json.loads(open('$OUT/bench_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['config'].get('step_form'))" || tail -5 "$OUT/bench_$m.err"; done;;
models) for m in ${MODELS:-dcnv2 din dssm}; do model_table $m > "$OUT/${m}_step_kernels.txt" 2>&1; head -3 "$OUT/${m}_step_kernels.txt"; done;;
ab) bash tools/r04_ab2.sh $OUT/ab ${AB_CASES} 2>&1 | tee "$OUT/ab.txt";;
hist) bash tools/r04_period.sh ${HIST_CASES:-ahead=} 2>&1 | tee "$OUT/period_hist.txt";;
*) echo "unknown stage $s";;
esac; done
echo "=== done $(date +%T)"
