mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-sweep --brief "${EXTRA[@]}" 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'], (d['config']['step_form'] or {}).get('chosen'))" || tail -5 gpurun_out/r3c/$tag.err; }
EXTRA=()
run ovl RECHUB_STEP_FORM=overlap RECHUB_SWEEP_GRID=512
run ovl_own RECHUB_STEP_FORM=overlap RECHUB_SWEEP_GRID=512 RECHUB_OWN_GEMM=1
run inline_own RECHUB_STEP_FORM=inline RECHUB_OWN_GEMM=1
for m in dcnv2 din dssm; do EXTRA=(--model $m --steps 50); run ${m}_auto RECHUB_X=0; done
