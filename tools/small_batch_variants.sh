#!/bin/bash
# Fused sweep variants against the materialising pipeline at small batches (one MI355X):
#   KPILQR_FUSED_WAVES=1 one wave per trajectory, 2 control/state split, 3 producer/consumer pair, 4 consumer/side/producer triple (backward)
#   KPILQR_FUSED_FWD_WAVES=1 one wave, 2 state/cost pair, 3 state/cost/staging triple (forward)
# "auto" = what the library picks by itself (pairs while 2 x batch <= #SIMDs).
for B in ${BATCHES:-1 128}; do
 echo -n "B=$B materialising            : "
 timeout -k 10 200 python bench.py --batch $B --unfused --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
 for W in ${FORMS:-"1 1" "2 1" "3 1" "3 2" "3 3" "4 3" "auto"}; do
   set -- $W
   if [ "$W" = "auto" ]; then unset KPILQR_FUSED_WAVES KPILQR_FUSED_FWD_WAVES; echo -n "B=$B fused auto              : ";
   else export KPILQR_FUSED_WAVES=$1 KPILQR_FUSED_FWD_WAVES=$2; echo -n "B=$B fused bwd_waves=$1 fwd_waves=$2 : "; fi
   timeout -k 10 200 python bench.py --batch $B --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stage_ms'].items()})"
 done
done
