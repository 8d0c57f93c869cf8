#!/bin/bash
set -e -o pipefail
out=$PWD/gpurun_out/${1:-ifs}
mkdir -p $out
for f in 16 20 24 32; do
  timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 10 --warmup 3 --inflight $f > $out/f$f.json 2> $out/f$f.err
  python3 -c "
import json;d=json.load(open('$out/f$f.json'));print('inflight $f', d['value'], d['plan_build_ms_per_batch'], d['setup_s'])"
done
