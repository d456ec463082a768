export TMPDIR=/tmp; mkdir -p gpurun_out/r03
(cd /tmp && rm -rf /tmp/din_prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/din_prof -o din -- python $OLDPWD/bench.py --model din --trace-inner --steps 10 --warmup 5 > /dev/null 2> $OLDPWD/gpurun_out/r03/din_prof.err)
python tools/prof_summary.py /tmp/din_prof > gpurun_out/r03/din_kernel_stats.txt 2>&1
python - <<'PY'
import csv,glob,os
f=max(glob.glob('/tmp/din_prof/**/*kernel_trace.csv',recursive=True),key=os.path.getsize)
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks=[i for i,r in enumerate(rows) if "batch_gather_kernel" in r[2]]
lo,hi=marks[-3],marks[-2]
agg={}
for st,en,n in rows[lo:hi]:
    k=n.replace("void ","").replace("(anonymous namespace)::","").replace("rechub::","").split("(")[0][:70]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(en-st)/1e3
print("step wall us", (rows[hi][0]-rows[lo][0])/1e3, "busy", sum(v[1] for v in agg.values()))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:32]: print("%-72s %3d %9.1f"%(k,v[0],v[1]))
PY
