export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -5 gpurun_out/r3f/$tag.err; }
EXTRA=()
run inline_g8192 RECHUB_STEP_FORM=inline
for g in 512 1024 2048; do run inline_g$g RECHUB_STEP_FORM=inline RECHUB_TUNE=2=$g; done
