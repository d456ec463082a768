# per-model kernel table of ONE traced step: bash tools/r03_model_stats.sh dcnv2 dssm
export TMPDIR=/tmp; mkdir -p gpurun_out/r03
for m in "$@"; do
(cd /tmp && rm -rf /tmp/${m}_prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/${m}_prof -o $m -- python $OLDPWD/bench.py --model $m --trace-inner --steps 10 --warmup 5 > /dev/null 2> $OLDPWD/gpurun_out/r03/${m}_prof.err)
python - $m <<'PY'
import csv,glob,os,sys
m=sys.argv[1]
f=max(glob.glob(f'/tmp/{m}_prof/**/*kernel_trace.csv',recursive=True),key=os.path.getsize)
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks=[i for i,r in enumerate(rows) if "batch_gather_kernel" in r[2]]
lo,hi=marks[-3],marks[-2]
agg={}
for st,en,n in rows[lo:hi]:
    k=n.replace("void ","").replace("(anonymous namespace)::","").replace("rechub::","").split("(")[0][:70]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=(en-st)/1e3
print(m, "step wall us", (rows[hi][0]-rows[lo][0])/1e3, "busy", sum(v[1] for v in agg.values()), "launches", sum(v[0] for v in agg.values()))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:26]: print("%-72s %3d %9.1f"%(k,v[0],v[1]))
PY
done
