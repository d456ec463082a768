mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --warmup 10 --no-cpu-baseline --no-kernel-sweep --brief "${EXTRA[@]}" 2> gpurun_out/r3c/$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/r3c/$tag.err; }
for i in 1 2 3; do EXTRA=(--steps 20); run tuned20_$i RECHUB_X=1; done
for i in 1 2; do EXTRA=(--steps 20); run inline20_$i RECHUB_SWEEP_OVERLAP=0; done
EXTRA=(--steps 200); run tuned200 RECHUB_X=1
