#!/bin/bash
# A/B of library builds at small batches (GPU box): AB_BATCHES="512 256 1" tools/ab_small.sh NAME...   (the in-tree build = "default")
for v in "$@"; do
  if [ "$v" = default ]; then unset KPILQR_LIB; else export KPILQR_LIB=$PWD/trajoptkp_amd/lib/variants/$v/libkpilqr.so; fi
  for B in ${AB_BATCHES:-512 256 1}; do
    python bench.py --workload-cache /tmp/kpwl --batch $B --no-secondary --no-cpu-baseline --steps 10 --warmup 2 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v B=$B', round(d['value']), d['stage_ms'], 'K %.1e cost %.1e' % (d['parity_check']['max_rel_err_K'], d['parity_check']['max_rel_err_cost_pred']))"
  done
done
