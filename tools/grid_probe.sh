#!/bin/bash
# rank-stage time for several persistent-grid sizes (tuning only): tools/grid_probe.sh
set -e -o pipefail
mkdir -p gpurun_out/grid
for g in 1024 1536 2048 3072 4096; do
  MGPU_RANK_GRID=$g RANK_SKIP=0,0 timeout -k 10 120 python3 tools/rank_parts.py > gpurun_out/grid/g$g.log 2>&1
  echo "grid $g: $(tail -n 1 gpurun_out/grid/g$g.log)"
done
