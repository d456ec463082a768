mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
cd /tmp
for cfg in "pad56k RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=57344"; do
  set -- $cfg; tag=$1; shift
  rm -rf /tmp/tl_$tag
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python $OLDPWD/bench.py --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> $OLDPWD/gpurun_out/r3c/tl_$tag.err
  python $OLDPWD/tools/timeline.py /tmp/tl_$tag 1 > $OLDPWD/gpurun_out/r3c/timeline_$tag.txt
  echo "== $tag"; cat $OLDPWD/gpurun_out/r3c/timeline_$tag.txt
done
cd $OLDPWD
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-sweep --brief 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -5 gpurun_out/r3c/$tag.err; }
for pad in 53248 59392 61440 63488; do run ovl_pad$pad RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=$pad; done
