#!/bin/bash
# Round 6, last call: every seed of the fuzz families (stride 1: ~850 random scenes) against the live oracle on the GPU box, both
# builds, with the large-frame stage forms forced (what the benchmark runs: list compactions, split pick) and with the forms small
# frames get by default.  Prints the tail of each leg: FUZZ {} = every tensor of every scene within the bars.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_soak; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MALLOC_MMAP_THRESHOLD_=1024 MALLOC_PERTURB_=255 PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests
for libm in exact default; do
  for forms in "RDR_LARGE_FRAME_FORMS=1" "X=0"; do
    echo "== build $libm, $forms"
    if [ $libm = exact ]; then export REDNER_AMD_LIBM=exact; else unset REDNER_AMD_LIBM; fi
    env $forms FUZZ_STRIDE=1 timeout 1500 python tests/test_fuzz_parity.py gpu 2>&1 | grep -E "^(FUZZ|SCENES|FLIPS|ORACLE_UNSTABLE) " | cut -c1-600
  done
done 2>&1 | tee $OUT/soak.txt
