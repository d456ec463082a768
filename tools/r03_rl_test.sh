export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -5 gpurun_out/r3f/$tag.err; }
EXTRA=()
for g in 256 320 384 448; do run pipe$g RECHUB_STEP_FORM=pipelined RECHUB_SWEEP_GRID=$g; done
run pipe512_own RECHUB_STEP_FORM=pipelined RECHUB_SWEEP_GRID=512 RECHUB_OWN_GEMM=1
run pipe384_own RECHUB_STEP_FORM=pipelined RECHUB_SWEEP_GRID=384 RECHUB_OWN_GEMM=1
