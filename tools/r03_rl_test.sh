export TMPDIR=/tmp; mkdir -p gpurun_out/r3g
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --brief 2>gpurun_out/r3g/$tag.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step']); print(json.dumps(d.get('gather_kernel_sweep') or d.get('roofline_north_star'))[:1500])" || tail -5 gpurun_out/r3g/$tag.err; }
run base RECHUB_X=1
run store RECHUB_TUNE=6=5
