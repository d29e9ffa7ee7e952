#!/bin/bash
# a4 / a6 inside the tiled sweeps, every combination, on BASELINE configs[2] and configs[4] (run on the GPU box):
#   bash tools/tiled_variants.sh > profiles/r02_tiled_a4_a6.txt
for cfg in "panda_pushing adaptive_jerk 3000 64 5" "high_dof_push iterative_error 5000 128 3"; do
  set -- $cfg
  for a4 in 0 1; do for a6 in 0 1; do
    KPILQR_TILED_A4=$a4 KPILQR_TILED_A6=$a6 python bench.py --task $1 --keypoints $2 --T $3 --batch $4 --steps $5 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 T=$3 B=$4 $2 KPILQR_TILED_A4=$a4 KPILQR_TILED_A6=$a6:', d['config']['kernels']['backward'], 'ms/iteration', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stage_ms'].items()}, 'K rel err', '%.1e' % d['parity_check']['max_rel_err_K'])"
  done; done
done
