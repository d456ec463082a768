mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-sweep --brief "${EXTRA[@]}" 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['rows_behind_after_flush'])" || tail -5 gpurun_out/r3c/$tag.err; }
for m in dcnv2 din; do EXTRA=(--model $m); for pad in 66000 90112 122880; do run ${m}_ovl_pad$pad RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=$pad; done; done
EXTRA=(--model dcnv2); run dcnv2_ovl_pad90k_g2048 RECHUB_SWEEP_OVERLAP=1 RECHUB_TUNE=3=90112,2=2048
