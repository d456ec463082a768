mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-kernel-sweep 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -3 gpurun_out/r3c/$tag.err; }
run base RECHUB_X=0
for n in 16 18 20 22; do run ovl_sw${n}_main$((32-n)) RECHUB_SWEEP_OVERLAP=1 RECHUB_SWEEP_CUS=$n RECHUB_MAIN_CUS=$((32-n)); done
run main16_only RECHUB_MAIN_CUS=16
run main24_only RECHUB_MAIN_CUS=24
