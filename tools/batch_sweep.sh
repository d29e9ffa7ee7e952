#!/bin/bash
# trajectory-iterations/s against the batch (GPU box, repo root): bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
# One line per batch: the bench line's value, ms per iteration and stage times (HIP events around each launch).
BATCHES=${BATCHES:-"1 8 32 64 128 256 384 512 768 1024 1536 2048 4096"}
echo "batch   trajectory-iterations/s   ms per iteration   fd_difference ms   backward ms   forward ms"
for B in $BATCHES; do
  timeout -k 10 300 python bench.py --batch $B --tiled-seeds --no-secondary --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stage_ms']
fd = '%17.3f' % s['fd_difference'] if 'fd_difference' in s else '         (inside)'
print('%5d %25s %18.3f  %s %13.3f %12.3f' % ($B, format(round(d['value']), ',').replace(',', ' '), d['ms_per_step'], fd, s['backward'], s['forward']))" || exit 1
done
