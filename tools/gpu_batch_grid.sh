#!/bin/bash
# backward time of the small-frame loop over (samples per batch) x (sample workers)
for cfg in "256 4" "256 16" "128 8" "512 4"; do
  set -- $cfg
  echo "== $1 x $1, $2 spp"
  for S in 1 2 4 8; do
    line="S=$S:"
    for W in 1 2 3 4; do
      t=$(RDR_BATCH=$S RDR_WORKERS=$W python tools/small_loop_timing.py $1 $2 2>/dev/null | grep backward | sed 's/.*median \([0-9.]*\) ms.*/\1/')
      line="$line  W$W $t"
    done
    echo "$line"
  done
done
