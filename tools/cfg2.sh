#!/bin/bash
# configs[2] (pushing n=20, B=64, T=3000, adaptive-jerk lists; CFG2_TASK=walker ... for the other two- and three-tile shapes) on library builds: tools/cfg2.sh NAME...   ("default" = in-tree)
# CFG2_RES=smooth|drawn (residual Jacobians; default both), CFG2_ARGS: extra bench.py arguments
for v in "$@"; do
  if [ "$v" = default ]; then unset KPILQR_LIB; else export KPILQR_LIB=$PWD/trajoptkp_amd/lib/variants/$v/libkpilqr.so; fi
  for res in ${CFG2_RES:-drawn smooth}; do
    if [ "$res" = smooth ]; then export KPILQR_BENCH_RESIDUALS=smooth; else unset KPILQR_BENCH_RESIDUALS; fi
    timeout -k 10 300 python bench.py --task ${CFG2_TASK:-panda_pushing} --keypoints ${CFG2_KP:-adaptive_jerk} --batch ${CFG2_B:-64} --T 3000 --tiled-seeds --no-secondary --no-cpu-baseline --steps 8 --warmup 2 $CFG2_ARGS 2>/dev/null > /tmp/cfg2.out
    grep "col wave\|wave " /tmp/cfg2.out | sort | uniq -c | sort -rn | head -8
    tail -1 /tmp/cfg2.out | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$res', round(d['value'],1), {k: round(x,2) for k,x in d['stage_ms'].items()}, d['parity_check']['pass'], d['parity_check']['max_rel_err_K'], d['config']['launched_backward'])"
  done
done
