export TMPDIR=/tmp; mkdir -p gpurun_out/r03
for m in ${MODELS:-dssm}; do
(cd /tmp && rm -rf /tmp/${m}_prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${m}_prof -o $m -- python $OLDPWD/bench.py --model $m --trace-inner --steps 10 --warmup 5 > /dev/null 2> $OLDPWD/gpurun_out/r03/${m}_prof.err)
python tools/timeline.py /tmp/${m}_prof 1 | grep -E "touched|seq_pool|embed_fwd|batch_gather" | cut -c1-140
done
(cd /tmp && rm -rf /tmp/rp_prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_prof -o rp -- python $OLDPWD/tools/refresh_probe.py > /tmp/rp.log 2>&1); head -3 /tmp/rp.log
python - <<'PY'
import csv,glob,os
f=max(glob.glob('/tmp/rp_prof/**/*kernel_trace.csv',recursive=True),key=os.path.getsize)
rows=[r for r in csv.DictReader(open(f)) if "adam_lazy_touched" in r["Kernel_Name"]]
for r in rows[:7]: print((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
PY
