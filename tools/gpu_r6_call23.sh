cd $GRAFT_REPO_ROOT
export REDNER_AMD_LIB=$GRAFT_REPO_ROOT/variants/gcount3.so
for w in living_room_standin_envmap living_room_standin; do
  echo "== $w"
  python bench.py --workload $w --spp 8 --steps 1 --warmup 0 --no-cpu-baseline --no-alone-leg --no-profile --no-self-check --no-large-leg 2>&1 | grep "\[gather\]" | sed 's/slots<= [0-9]* //' | sort | uniq -c | sort -rn | head -14
done
