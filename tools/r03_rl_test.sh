export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --timeout 300 -k "graph_mode or match or dssm or din or dien" 2>&1 | tail -4 | cut -c1-220
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3f/$tag.err; }
EXTRA=(--model dssm); run dssm_ahead RECHUB_X=1; run dssm_noahead RECHUB_REFRESH_AHEAD=0
