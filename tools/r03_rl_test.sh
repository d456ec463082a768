export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
ROOT=$PWD
(cd /tmp && rm -rf /tmp/tl_new && cd $ROOT && RECHUB_OWN_GEMM=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_new -o t -- python bench.py --trace-inner --steps 30 --warmup 10 > /dev/null 2> $ROOT/gpurun_out/r3g/tl_new.err)
python tools/timeline.py /tmp/tl_new 1 | cut -c1-120
