#!/bin/bash
# Profiles of the round (run on the GPU box from the repo root: `gpurun -- bash tools/collect_profiles.sh r04 [per_step]`):
#   gpurun_out/<tag>/bench.json            the default bench line
#   gpurun_out/<tag>/stats/                rocprofv3 --kernel-trace --stats of the same command (no side measurements)
#   gpurun_out/<tag>/pmc_{fetch,write,m3,m4}/  separate --pmc passes (HBM traffic, MFMA / VALU / LDS counters)
#   gpurun_out/<tag>/generic.json          the VALU/LDS (non-MFMA) kernels at the same workload: evidence for MFMA at n = 14
# tools/pmc_summarise.py turns the counter CSVs into profiles/<tag>_pmc_*.json.
# Second argument "per_step": the same passes (without the generic ones) for `bench.py --streamed-jacobians` -- the residual
# Jacobians given per step, the round-1..3 form -- into profiles/<tag>_*_per_step_jacobians.*
set -o pipefail
TAG=${1:-r05}
MODE=${2:-}
EXTRA=""
SUB=$TAG
if [ "$MODE" = per_step ]; then EXTRA="--streamed-jacobians"; SUB=${TAG}_per_step; fi
OUT=$PWD/gpurun_out/$SUB
mkdir -p $OUT
export TMPDIR=/tmp
# the 1024-distinct-seed workload is generated once (forked workers, before HIP starts) and memory-mapped by every later pass:
# under rocprofv3 the profiler has initialised the GPU before the script starts, so those passes must not fork
WL="--workload-cache /tmp/kpilqr_workload"
if [ "$MODE" = per_step ]; then
  python bench.py $WL $EXTRA --no-secondary --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err || exit 1
else
  python bench.py $WL --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err || exit 1
fi
echo "bench done"
if [ "$MODE" != per_step ]; then
  python bench.py $WL --generic --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/generic.json 2> $OUT/generic.err || exit 1
  echo "generic done"
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $OLDPWD/bench.py $WL $EXTRA --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $OUT/stats.log 2>&1 || exit 1
echo "stats done"
if [ "$MODE" != per_step ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_generic -- python3 $OLDPWD/bench.py $WL --generic --no-secondary --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats_generic.log 2>&1 || exit 1
  echo "generic stats done"
fi
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "m3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU" "m4 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  set -- $pass; name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $OLDPWD/bench.py $WL $EXTRA --no-secondary --no-cpu-baseline --steps 3 --warmup 1 > $OUT/pmc_$name.log 2>&1 || exit 1
  echo "pmc $name done"
done
if [ "$MODE" != per_step ]; then
  # the headline shape with the reference's own key-point method for reaching (per-DoF lists: reaching.yaml:6-8) and adaptive_jerk
  for kp in reach_velocity_change reach_adaptive_jerk; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$kp -- python3 $OLDPWD/bench.py --keypoints $kp --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $OUT/stats_$kp.log 2>&1 || exit 1
    echo "stats $kp done"
  done
  # a GPU's share of the 1024 trajectories on 2 / 4 GPUs: the consumer / helper pair and the state / cost pair.  The workload cache
  # is keyed by the batch: warm it with a plain run first -- under rocprofv3 the profiler has initialised the GPU before the script
  # starts and a cache miss there would fork the generator's worker pool behind it (round-4 advisor)
  for bb in 512 256; do
    python3 $OLDPWD/bench.py $WL --batch $bb --no-secondary --no-cpu-baseline --steps 2 --warmup 1 --detail $OUT/warm_b$bb.json > $OUT/warm_b$bb.log 2>&1 || { echo "warm-up run B=$bb failed"; tail -5 $OUT/warm_b$bb.log; exit 1; }
  done
  export KPILQR_BENCH_NOFORK=1
  for bb in 512 256; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b$bb -- python3 $OLDPWD/bench.py $WL --batch $bb --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $OUT/stats_b$bb.log 2>&1 || exit 1
    echo "stats batch $bb done"
  done
fi
if [ "$MODE" != per_step ]; then
  # the tiled shapes of BASELINE configs[2] and configs[4] (8 / 2 seeds tiled: nothing forks)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs2 -- python3 $OLDPWD/bench.py --task panda_pushing --keypoints adaptive_jerk --batch 64 --T 3000 --tiled-seeds --no-secondary --no-cpu-baseline --steps 8 --warmup 2 > $OUT/stats_configs2.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_configs4 -- python3 $OLDPWD/bench.py --task high_dof_push --keypoints iterative_error --batch 128 --T 5000 --tiled-seeds --no-secondary --no-cpu-baseline --steps 4 --warmup 1 > $OUT/stats_configs4.log 2>&1 || exit 1
  echo "stats configs[2], configs[4] done"
fi
cd $OLDPWD
python tools/pmc_summarise.py $OUT $TAG $MODE || exit 1
echo "summaries written"
