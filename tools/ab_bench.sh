#!/bin/bash
# A/B of library builds on the headline workload (GPU box): tools/ab_bench.sh NAME...   (the in-tree build = "default")
# AB_ARGS: extra bench.py arguments (e.g. AB_ARGS="--keypoints reach_adaptive_jerk" for per-DoF key-point lists)
for v in "$@"; do
  if [ "$v" = default ]; then unset KPILQR_LIB; else export KPILQR_LIB=$PWD/trajoptkp_amd/lib/variants/$v/libkpilqr.so; fi
  python tools/ab_check.py 2>&1 | tail -1
  python bench.py --workload-cache /tmp/kpwl --no-secondary --no-cpu-baseline --steps 10 --warmup 2 $AB_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), d['stage_ms'], d['parity_check']['max_rel_err_K'])"
done
