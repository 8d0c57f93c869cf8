#!/bin/bash
# throughput against the rank kernel's persistent grid and the batches in flight: tools/grid_sweep.sh <tag>
set -e -o pipefail
out=$PWD/gpurun_out/${1:-gs}
mkdir -p $out
for g in 1024 1536 2048 3072 4096; do
  MGPU_RANK_GRID=$g timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 6 --warmup 2 > $out/g$g.json 2> $out/g$g.err
  python3 -c "
import json;d=json.load(open('$out/g$g.json'));print('rank grid $g', d['value'], d['stages']['rank']['ms'])"
done
for f in 6 10 12 16; do
  timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 6 --warmup 2 --inflight $f > $out/f$f.json 2> $out/f$f.err
  python3 -c "
import json;d=json.load(open('$out/f$f.json'));print('inflight $f', d['value'])"
done
