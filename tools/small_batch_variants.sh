#!/bin/bash
# Fused sweep variants against the materialising pipeline at small batches (one MI355X):
#   KPILQR_FUSED_WAVES=1 one wave per trajectory, 2 control/state split, 3 producer/consumer pair (backward)
#   KPILQR_FUSED_FWD_WAVES=1 one wave, 2 state/cost pair (forward)
for B in ${BATCHES:-1 128}; do
 for MODE in "--unfused" "--fused" ; do
  for W in "1 1" "2 1" "3 1" "3 2"; do
   set -- $W
   if [ "$MODE" = "--unfused" ] && [ "$W" != "1 1" ]; then continue; fi
   echo -n "B=$B $MODE bwd_waves=$1 fwd_waves=$2 : "
   KPILQR_FUSED_WAVES=$1 KPILQR_FUSED_FWD_WAVES=$2 timeout -k 10 200 python bench.py --batch $B $MODE --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
  done
 done
done
