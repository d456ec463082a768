export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
timeout 200 python tools/sweep_alone_probe.py 2>&1 | grep -E " 512| 8192| 256 "
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -q -x -k "flush_leaves_no_row_behind or lazy" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3g/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3g/$tag.err; }
EXTRA=(); run deepfm X=1; run deepfm X=1
EXTRA=(--model dssm); run dssm X=1
