#!/bin/bash
# A/B of two builds of libmarlin_gpu.so on ONE box: tools/ab_libs.sh <tag> <other.so> [bench args]
set -e -o pipefail
out=$PWD/gpurun_out/${1:-ab}; other=$2; shift 2
mkdir -p $out
L=marlin_amd/lib
cp $L/libmarlin_gpu.so $out/this.so
for rep in 1 2; do
  for v in this other; do
    if [ $v = other ]; then cp $other $L/libmarlin_gpu.so; else cp $out/this.so $L/libmarlin_gpu.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 6 --warmup 2 "$@" > $out/$v$rep.json 2> $out/$v$rep.err || { cp $out/this.so $L/libmarlin_gpu.so; tail -3 $out/$v$rep.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$out/$v$rep.json'));s=d['stages'];print('$v$rep', d['value'], 'rank', s['rank']['ms'], 'select', s['select']['ms'], 'formula', s['formula+compact']['ms'], 'facets', s['facets+aggs']['ms'])"
  done
done
cp $out/this.so $L/libmarlin_gpu.so; rm -f $out/this.so
