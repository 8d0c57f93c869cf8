#!/bin/bash
# bench of the rank stage with the record-layout switches (read at mirror build): tools/rank_variants.sh <tag>
set -e -o pipefail
out=$PWD/gpurun_out/${1:-rv}
mkdir -p $out
for v in default NO_RECLEN NO_POOLIN BOTH; do
  case $v in
    default) e="";; NO_RECLEN) e="MGPU_NO_RECLEN=1";; NO_POOLIN) e="MGPU_NO_POOLIN=1";; BOTH) e="MGPU_NO_RECLEN=1 MGPU_NO_POOLIN=1";;
  esac
  env $e timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 6 --warmup 2 > $out/$v.json 2> $out/$v.err
  python3 -c "
import json,sys;d=json.load(open('$out/$v.json'));print('$v', d['value'], d['stages']['rank'])"
done
