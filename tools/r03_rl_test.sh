export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -x -k "refresh or lazy or padding or flush" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3g/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3g/$tag.err; }
EXTRA=(); run deepfm X=1; run deepfm X=1
EXTRA=(--model dssm); run dssm X=1
EXTRA=(--model dcnv2); run dcnv2 X=1
(cd /tmp && rm -rf /tmp/tl_new && cd $ROOT && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_new -o t -- python bench.py --trace-inner --steps 30 --warmup 10 > /dev/null 2> $ROOT/gpurun_out/r3g/tl_new.err)
python tools/timeline.py /tmp/tl_new 1 | head -5 | cut -c1-120
