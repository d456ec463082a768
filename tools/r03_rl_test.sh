export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config']['step_form']['chosen'])" || tail -5 gpurun_out/r3f/$tag.err; }
EXTRA=(); run deepfm RECHUB_X=1
EXTRA=(--model dcnv2); run dcnv2 RECHUB_X=1
EXTRA=(--model din); run din RECHUB_X=1; run din_deferred RECHUB_STEP_FORM=deferred RECHUB_SWEEP_GRID=256
MODELS=dssm bash tools/r03_model_prof.sh 2>&1 | grep -vE "bn_|Cijk|wgrad|l2norm" | tail -32
