mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu -k "graph_mode_flush" --timeout 200 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_properties.py -q -m gpu -k "lazy_adam_equals" --timeout 200 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-sweep --brief 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -5 gpurun_out/r3c/$tag.err; }
run base RECHUB_X=0
run ovl RECHUB_SWEEP_OVERLAP=1
run ovl_pad56k RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=57344
run ovl_pad58k RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=59392
