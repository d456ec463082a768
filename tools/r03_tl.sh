mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
cd /tmp
for cfg in "branch RECHUB_STEP_FORM=branch RECHUB_SWEEP_GRID=512"; do
  set -- $cfg; tag=$1; shift
  rm -rf /tmp/tl_$tag
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python $OLDPWD/bench.py --trace-inner --steps 30 --warmup 10 --rows 4000000 > /dev/null 2> $OLDPWD/gpurun_out/r3c/tl_$tag.err
  python $OLDPWD/tools/timeline.py /tmp/tl_$tag 1 > $OLDPWD/gpurun_out/r3c/timeline_$tag.txt
  echo "== $tag"; cat $OLDPWD/gpurun_out/r3c/timeline_$tag.txt
done
