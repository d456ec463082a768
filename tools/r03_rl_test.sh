export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu --timeout 300 -k "graph_mode_flush_leaves_no_row_behind_ctr" -x 2>&1 | tail -12 | cut -c1-220
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>gpurun_out/r3f/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'], d['config']['step_form'])" || tail -5 gpurun_out/r3f/$tag.err; }
EXTRA=()
for g in 384 448 512 640; do run branch$g RECHUB_STEP_FORM=branch RECHUB_SWEEP_GRID=$g; done
run defer512 RECHUB_STEP_FORM=deferred RECHUB_SWEEP_GRID=512
