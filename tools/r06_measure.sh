#!/bin/bash
# Round-6 measurement session (one gpurun call; STAGES selects parts).  Outputs under gpurun_out/r06${TAG:+_$TAG}/ (the
# summaries that are cited are copied into profiles/ by hand).
#   integ      mounted-reference integration tests on the GPU (tools/mount_reference.sh first, in the build container)
#   tests      the driver's command: pytest tests -m gpu -x -q
#   testsel    pytest -m gpu on TEST_SEL (file list), TEST_K = a -k expression (may hold spaces)
#   bench      the driver's bench command (incl. CPU baseline, dense twin check)
#   bench300   300 steps, no CPU baseline (steady state)
#   timeline   one traced step of the default workload (tools/timeline.py)
#   prof       rocprofv3 kernel stats of the driver's command
#   dp         one-rank data-parallel step, both placements
#   dptrace    per-kernel table of one traced data-parallel step
#   models     per-kernel tables of one traced step of MODELS (default dcnv2 din dssm)
#   modelbench bench.py --model m for MODELS
#   ab         same-box A/B lines: AB_CASES="name|ENV=..;ENV2=..|bench args" ...
#   hist       step period distribution without a profiler (tools/period_hist.py)
#   crash      tools/bitwise_probe.py dp under faulthandler and under rocgdb (backtrace of the capture segfault)
#   ceiling    tools/bwd_ceiling_probe.py (timing) + its two PMC passes
#   cmd        run CMD verbatim
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06${TAG:+_$TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
STAGES=${STAGES:-"tests bench"}
MODELS=${MODELS:-"dcnv2 din dssm"}
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
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:40]: print("%-72s %3d %9.1f"%(k,v[0],v[1]))
PY
}
brief() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","ms_per_step_steady","flush_ms","rows_behind_after_flush")}, (d.get("config") or {}).get("step_form"))
except Exception as e:
    print("FAILED", e)
PY
}
for s in $STAGES; do echo "=== stage $s $(date +%T)"; case $s in
integ) RECHUB_REFERENCE=$PWD/_refmount timeout 900 python -m pytest -m gpu tests/test_integration_patch.py -v > "$OUT/integration_gpu.log" 2>&1; echo "rc=$?"; tail -15 "$OUT/integration_gpu.log";;
tests) timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?"; tail -5 "$OUT/pytest_gpu.log";;
testsel) timeout 1200 python -m pytest -m gpu -x -q ${TEST_SEL:-tests} ${TEST_K:+-k "$TEST_K"} > "$OUT/pytest_sel${SELTAG:-}.log" 2>&1; echo "rc=$?"; tail -15 "$OUT/pytest_sel${SELTAG:-}.log";;
bench) ( time timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real
  brief "$OUT/bench_default.json"; tail -5 "$OUT/bench_default.err";;
bench300) timeout 400 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief ${BENCH300_ARGS:-} > "$OUT/bench_300.json" 2> "$OUT/bench_300.err"; echo "rc=$?"; brief "$OUT/bench_300.json";;
timeline) (cd /tmp && rm -rf /tmp/tl_r06 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_r06 -o t -- python "$OLDPWD/bench.py" --trace-inner --steps 30 --warmup 10 --rows 4000000 ${TL_ARGS:-} > /dev/null 2> "$OLDPWD/$OUT/tl.err"); python tools/timeline.py /tmp/tl_r06 2 > "$OUT/step_timeline${TLTAG:-}.txt" 2>&1; head -40 "$OUT/step_timeline${TLTAG:-}.txt";;
prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --brief --no-twin-check > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err"); echo "rc=$?"
  python tools/prof_summary.py "$OUT/prof" > "$OUT/prof_summary.txt" 2>&1; head -30 "$OUT/prof_summary.txt"
  cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
  find "$OUT/prof" -name '*kernel_trace.csv' -size +20M -delete;;
dp) for pl in replicate shard; do
    timeout 400 python bench.py --force-dp --tables $pl --steps 100 --warmup 10 --no-cpu-baseline --brief > "$OUT/bench_dp_$pl.json" 2> "$OUT/bench_dp_$pl.err"; echo "dp $pl rc=$?"
    brief "$OUT/bench_dp_$pl.json" || tail -5 "$OUT/bench_dp_$pl.err"; done;;
dptrace) for pl in ${DP_PLACEMENTS:-shard replicate}; do
  (cd /tmp && rm -rf /tmp/dp_prof_$pl && MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dp_prof_$pl -o dp -- python $OLDPWD/bench.py --force-dp --tables $pl --trace-inner --steps 10 --warmup 5 --rows 4000000 ${DP_ARGS:-} > /dev/null 2> $OLDPWD/$OUT/dp_prof_$pl.err)
  python tools/timeline.py /tmp/dp_prof_$pl 2 > "$OUT/dp_${pl}_step_timeline.txt" 2>&1; head -45 "$OUT/dp_${pl}_step_timeline.txt"; done;;
models) for m in $MODELS; do model_table $m > "$OUT/${m}_step_kernels.txt" 2>&1; head -45 "$OUT/${m}_step_kernels.txt"; done;;
modelbench) for m in $MODELS; do timeout 300 python bench.py --model $m --steps 100 --warmup 10 --no-cpu-baseline --brief > "$OUT/bench_$m.json" 2> "$OUT/bench_$m.err"; echo "$m rc=$?"; brief "$OUT/bench_$m.json"; done;;
ab) IFS=' ' read -r -a cases <<< "${AB_CASES:-base||}"
  for rep in $(seq 1 ${AB_REPEAT:-2}); do for c in "${cases[@]}"; do
    name=${c%%|*}; rest=${c#*|}; envs=${rest%%|*}; args=${rest#*|}; args=${args//;/ }  # (bench args: ';' for ' ')
    ( IFS=';'; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS
      timeout 300 python bench.py --steps ${AB_STEPS:-200} --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --no-step-accounting --no-pmc --no-twin-check $args > "$OUT/ab_${name}_$rep.json" 2> "$OUT/ab_${name}_$rep.err" )
    python - "$OUT/ab_${name}_$rep.json" "$name" "$rep" <<'PY' | tee -a "$OUT/ab_table.txt"
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-28s run %s  ms_per_step %.4f  value %.0f  form %s"%(sys.argv[2],sys.argv[3],d["ms_per_step"],d["value"],(d.get("config") or {}).get("step_form")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done; done;;
hist) timeout 300 python tools/period_hist.py ${HIST_ARGS:-} > "$OUT/period_hist${HISTTAG:-}.txt" 2>&1; tail -12 "$OUT/period_hist${HISTTAG:-}.txt";;
crash) timeout 400 python -X faulthandler tools/bitwise_probe.py dp > "$OUT/crash_probe.txt" 2> "$OUT/crash_probe.err"; echo "rc=$?"; tail -5 "$OUT/crash_probe.txt"; tail -40 "$OUT/crash_probe.err"
  if [ -x /opt/rocm/bin/rocgdb ]; then timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop" -ex run -ex "bt 40" -ex "info threads" --args python tools/bitwise_probe.py dp > "$OUT/crash_gdb.txt" 2>&1; echo "gdb rc=$?"; grep -n "SIGSEGV" -A 60 "$OUT/crash_gdb.txt" | head -120; fi;;
ceiling) timeout 600 python tools/bwd_ceiling_probe.py ${CEIL_ARGS:-} > "$OUT/bwd_ceiling.txt" 2>&1; echo "rc=$?"; cat "$OUT/bwd_ceiling.txt"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rm -rf /tmp/ceil_$c && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ceil_$c -o c -- python "$OLDPWD/tools/bwd_ceiling_probe.py" --pmc --batches 65536 > /dev/null 2> "$OLDPWD/$OUT/ceil_$c.err"); echo "pmc $c rc=$?"
  done
  python tools/bwd_ceiling_pmc.py /tmp/ceil_FETCH_SIZE /tmp/ceil_WRITE_SIZE 8 > "$OUT/bwd_ceiling_pmc.txt" 2>&1; cat "$OUT/bwd_ceiling_pmc.txt";;
cmd) bash -c "${CMD}" > "$OUT/cmd${CMDTAG:-}.log" 2>&1; echo "rc=$?"; tail -${CMDTAIL:-40} "$OUT/cmd${CMDTAG:-}.log";;
*) echo "unknown stage $s";;
esac; done
echo "=== done $(date +%T)"
