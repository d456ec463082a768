export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu --timeout 300 -k "prelu or dssm or match or mlp" 2>&1 | tail -12 | cut -c1-220
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep --model dssm 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])" || tail -5 gpurun_out/r3f/$tag.err; }
run dssm_fused RECHUB_X=1
run dssm_unfused RECHUB_BN_PRELU=0
