export TMPDIR=/tmp; mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --timeout 300 -k "graph_mode" 2>&1 | tail -3 | cut -c1-200
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --brief 2>gpurun_out/r3f/gs.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('gather_kernel_sweep') or {}; print(d['ms_per_step']); [print(k, {n:(v[n]['avg_ms'],v[n]['frac']) for n in ('rh_embed_fwd','rh_embed_bwd','rh_embed_bwd_rows') if n in v}) for k,v in g.items()]" || tail -5 gpurun_out/r3f/gs.err
