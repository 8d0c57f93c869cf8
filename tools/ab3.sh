#!/bin/bash
# several builds of libmarlin_gpu.so on ONE box: tools/ab3.sh <tag> <a.so> <b.so> ...  (the tree's own build is "this")
set -e -o pipefail
out=$PWD/gpurun_out/${1:-ab3}; shift
mkdir -p $out
L=marlin_amd/lib
cp $L/libmarlin_gpu.so $out/this.so
for rep in 1 2; do
  for v in this "$@"; do
    if [ $v = this ]; then cp $out/this.so $L/libmarlin_gpu.so; else cp $v $L/libmarlin_gpu.so; fi
    n=$(basename $v .so)
    timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 6 --warmup 2 > $out/$n$rep.json 2> $out/$n$rep.err || { cp $out/this.so $L/libmarlin_gpu.so; tail -3 $out/$n$rep.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$out/$n$rep.json'));s=d['stages'];print('$n$rep', d['value'], 'rank', s['rank']['ms'], 'select', s['select']['ms'], 'formula', s['formula+compact']['ms'], 'facets', s['facets+aggs']['ms'])"
  done
done
cp $out/this.so $L/libmarlin_gpu.so; rm -f $out/this.so
