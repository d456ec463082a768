export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3g/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3g/$tag.err; }
EXTRA=(--model dssm); run dssm X=1; run dssm X=1
EXTRA=(); run deepfm X=1
