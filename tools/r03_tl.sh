# kernel timeline of one traced step: bash tools/r03_tl.sh MODEL [ENV=VALUE ...]
mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
m=$1; shift
cd /tmp; rm -rf /tmp/tl_$m
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$m -o t -- python $OLDPWD/bench.py --model $m --trace-inner --steps 10 --warmup 5 > /dev/null 2> $OLDPWD/gpurun_out/r3c/tl_$m.err
python $OLDPWD/tools/timeline.py /tmp/tl_$m 1 > $OLDPWD/gpurun_out/r3c/timeline_$m.txt
cat $OLDPWD/gpurun_out/r3c/timeline_$m.txt
