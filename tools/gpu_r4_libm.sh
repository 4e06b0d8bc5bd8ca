#!/bin/bash
# Round 4, last GPU call: the kernels with glibc's transcendental functions (csrc/libm_exact.h).
# 1. the whole GPU suite (new: test_libm_exact_gpu, fisheye / panorama fixtures with secondary edges sample-exact)
# 2. A/B on this box: the shipped build against variants/platform_libm.so (the device's own libm, rounds 1-3)
# 3. the bench line of the shipped build
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_libm
rm -rf $OUT; mkdir -p $OUT
export RDR_PARITY_REPORT=$OUT/parity_report.jsonl
timeout 420 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > $OUT/pytest.log
unset RDR_PARITY_REPORT
tail -15 $OUT/pytest.log
Q="--steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-self-check --no-alone-leg"
for round in 1 2; do
  for lib in redner_amd/lib/libredner_amd.so variants/platform_libm.so; do
    REDNER_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python bench.py $Q 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), 'Msamples/s  frac', round(d['roofline']['frac'],3))" | tee -a $OUT/ab_libm.txt
  done
done
timeout 200 python bench.py --steps 2 --no-cpu-baseline --no-profile 2> $OUT/bench.err | tail -1 > $OUT/bench_libm_exact.json
cut -c1-400 $OUT/bench_libm_exact.json
for w in living_room_standin living_room_standin_envmap; do
  timeout 200 python bench.py --workload $w --spp 64 --steps 2 --no-cpu-baseline --no-profile --no-self-check 2> /dev/null | tail -1 > $OUT/bench_$w.json
  python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read()); print('$w', round(d['value'],2))"
done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
