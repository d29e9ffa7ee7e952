#!/bin/bash
# A/B of the tiled sweeps with and without the round-3 changes (GPU box, repo root): bash tools/tiled_uw_ab.sh > gpurun_out/tiled_uw.txt
#   KPILQR_TILED_UW=0: the round-2 kernels (column decomposition, refresh replicated in every wave, run-time chunk counts)
#   KPILQR_TILED_UW=1: u-wave sweep for two / three tiles, interleaved chains for four tiles and in the forward kernel
# and, when a -DKP_CYC_UW -DKP_CYC_FT build of the library is given as $1, the per-step cycle counters the kernels print.
run() {  # label task keypoints batch T
  for uw in 0 1 0 1; do
    KPILQR_TILED_UW=$uw timeout -k 10 300 python bench.py --task $2 --keypoints $3 --batch $4 --T $5 --tiled-seeds --no-secondary --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stage_ms']; pc = d['parity_check']
print('%-34s UW=%s  %8.0f trajectory-iterations/s  backward %7.3f ms  forward %7.3f ms  (%s / %s; K within %.1e of the oracle: %s)' % ('$1', '$uw', d['value'], s['backward'], s['forward'], d['config']['kernels']['backward'], d['config']['kernels']['forward'], pc['max_rel_err_K'], 'pass' if pc['pass'] else 'FAIL'))" || exit 1
  done
}
run "pushing n=20 (2 tiles) B=64 T=3000" panda_pushing adaptive_jerk 64 3000
run "walker n=18 m=6 (2 tiles) B=64 T=3000" walker set_interval 64 3000
run "light clutter n=38 (3 tiles) B=64 T=3000" light_clutter_push set_interval 64 3000
run "high-DoF n=62 (4 tiles) B=128 T=5000" high_dof_push iterative_error 128 5000
if [ -n "$1" ]; then
  echo; echo "cycle counters per step (library built with -DKP_CYC_UW -DKP_CYC_FT; each probe costs ~100 cycles), pushing n=20, trajectory 0:"
  KPILQR_LIB=$1 timeout -k 10 200 python bench.py --task panda_pushing --keypoints adaptive_jerk --batch 64 --tiled-seeds --no-secondary --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | grep "wave" | sort -u
fi
