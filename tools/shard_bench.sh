cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_models.py -q -m gpu -k "data_parallel_machinery" > gpurun_out/t3.log 2>&1; grep -E "^E  |passed|failed|Error" gpurun_out/t3.log | head -30
for cfg in "0 replicate" "0 shard" "2 replicate" "2 shard" "8 replicate" "8 shard"; do
  set -- $cfg
  RECHUB_EMULATE_WORLD=$1 timeout 300 python bench.py --steps 300 --warmup 60 --graph 1 --no-cpu-baseline --force-dp --tables $2 2>gpurun_out/sb_$1_$2.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('emu=$1 tables=$2', d['value'], d['ms_per_step'], d['config']['hipgraph'], {n:round(v['avg_ms'],4) for n,v in k.items() if v.get('avg_ms')})"
done
