#!/bin/bash
# Backward wave forms of the fused sweep (GPU box): tools/pair_forms.sh [batch...]
# PF_WAVES: the KPILQR_FUSED_WAVES values to compare (default "3 5": producer / consumer pair against consumer / helper pair; 4 = triple)
# PF_ARGS: extra bench.py arguments (e.g. --streamed-jacobians); PF_RAW: the KPILQR_FUSED_RAW values to run (default "1 0")
for B in ${@:-512 384 320}; do
  for W in ${PF_WAVES:-3 5}; do for RAW in ${PF_RAW:-1 0}; do
    KPILQR_FUSED_WAVES=$W KPILQR_FUSED_RAW=$RAW python bench.py --workload-cache /tmp/kpwl --batch $B --no-secondary --no-cpu-baseline --steps 10 --warmup 2 $PF_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B waves=$W raw=$RAW', round(d['value']), d['stage_ms'], d['config']['launched']['backward'], 'K err %.1e' % d['parity_check']['max_rel_err_K'])"
  done; done
done
