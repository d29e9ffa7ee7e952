#!/bin/bash
# Forward wave forms of the fused sweep (GPU box): tools/fwd_forms.sh [batch...]   KPILQR_FUSED_FWD_WAVES = 1 one wave | 2 state / cost pair | 3 triple
for B in ${@:-512 256 64 1}; do
  for W in ${FF_WAVES:-1 2 3}; do
    if [ $W = 3 ] && [ $B -gt 256 ]; then continue; fi
    KPILQR_FUSED_FWD_WAVES=$W python bench.py --workload-cache /tmp/kpwl --batch $B --no-secondary --no-cpu-baseline --steps 10 --warmup 2 $FF_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B fwd_waves=$W', round(d['value']), d['stage_ms'], d['config']['launched']['forward'], 'cost err %.1e' % d['parity_check']['max_rel_err_cost_pred'])"
  done
done
