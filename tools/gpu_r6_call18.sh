#!/bin/bash
# Round 6, call 18: the last-bounce emitter test in the fused bounce form (small frames): fixtures on both builds, small optimisation
# loops of the previous library (variants/prev.so) against the in-tree one, the benchmark job once each.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_r; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_config_parity.py tests/test_backward_parity.py tests/test_forward_parity.py tests/test_default_library_gpu.py tests/test_sample_batches.py tests/test_fuzz_parity.py tests/test_tuning.py tests/test_edge_cases.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_subset.log
for rep in 1 2 3; do
  for spec in "previous|REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/prev.so" "fused form with the emitter test|X=0"; do
    label=${spec%%|*}; envs=${spec#*|}
    for cfg in "256 4" "128 8" "512 4"; do echo -n "$label [$cfg]: "; env $envs python tools/small_loop_timing.py $cfg 2>&1 | grep iteration; done
  done
done | tee $OUT/small_loop_ab.txt
tools/gpu_r6_exp.sh "previous commit|REDNER_AMD_LIB=variants/prev.so" "in-tree|X=0"
