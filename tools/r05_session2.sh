set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
OUT=gpurun_out/r05_wg; mkdir -p $OUT
echo "=== tests $(date +%T)"
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -x -q -k "grouped_weight or mlp_chain or full_size_graph_step or device_loader_training_equals_host_loader_training_bitwise" > $OUT/pytest_sel.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_sel.log
echo "=== tests with the switch on $(date +%T)"
RECHUB_AB=chainwgroup=0 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "full_size_graph_step or flush_leaves or data_parallel_machinery_on_one_rank_equals_plain_training_bitwise" > $OUT/pytest_sel_on.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_sel_on.log
echo "=== ab $(date +%T)"
bash tools/r04_ab2.sh $OUT/ab "base_1|RECHUB_AB=chainwgroup=0|" "wgroup_1||" "base_2|RECHUB_AB=chainwgroup=0|" "wgroup_2||" "k96||--lazy-k 96" "k160||--lazy-k 160" "grid448|RECHUB_SWEEP_GRID=448;RECHUB_STEP_FORM=deferred|" "grid640|RECHUB_SWEEP_GRID=640;RECHUB_STEP_FORM=deferred|" 2>&1 | tee $OUT/ab.txt
echo "=== done $(date +%T)"
