#!/bin/bash
# Fused backward variants against the materialising pipeline at small batches (one MI355X):
#   KPILQR_FUSED_WAVES=1 one wave per trajectory, 2 control/state split, 3 producer/consumer pair
for B in ${BATCHES:-1 128}; do
 for MODE in "--unfused" "--fused" ; do
  for W in 1 2 3; do
   if [ "$MODE" = "--unfused" ] && [ $W != 1 ]; then continue; fi
   echo -n "B=$B $MODE waves=$W : "
   KPILQR_FUSED_WAVES=$W timeout -k 10 200 python bench.py --batch $B $MODE --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
  done
 done
done
