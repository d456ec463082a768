export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "din or dien or bst" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3g/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3g/$tag.err; }
EXTRA=(--model din); run din RECHUB_X=1
