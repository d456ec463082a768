export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --brief --no-kernel-sweep "${EXTRA[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['flush_ms'])"; }
for k in 64 80 96 112 64 96; do EXTRA=(--lazy-k $k); run k$k RECHUB_STEP_FORM=overlap RECHUB_SWEEP_GRID=512; done
