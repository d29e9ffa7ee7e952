#!/bin/bash
# configs[4] (n=62, B=128, T=5000, iterative-error lists) on library builds: tools/cfg4.sh NAME...   ("default" = in-tree)
# CFG4_RES=smooth|drawn (residual Jacobians; default both), CFG4_ARGS: extra bench.py arguments
for v in "$@"; do
  if [ "$v" = default ]; then unset KPILQR_LIB; else export KPILQR_LIB=$PWD/trajoptkp_amd/lib/variants/$v/libkpilqr.so; fi
  for res in ${CFG4_RES:-drawn smooth}; do
    if [ "$res" = smooth ]; then export KPILQR_BENCH_RESIDUALS=smooth; else unset KPILQR_BENCH_RESIDUALS; fi
    timeout -k 10 300 python bench.py --task high_dof_push --keypoints iterative_error --batch 128 --T 5000 --tiled-seeds --no-secondary --no-cpu-baseline --steps 4 --warmup 1 $CFG4_ARGS 2>/dev/null > /tmp/cfg4.out
    grep "col wave\|wave " /tmp/cfg4.out | sort | uniq -c | sort -rn | head -8
    tail -1 /tmp/cfg4.out | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$res', round(d['value'],1), {k: round(x,2) for k,x in d['stage_ms'].items()}, d['parity_check']['pass'], d['parity_check']['max_rel_err_K'], d['config']['launched_backward'])"
  done
done
