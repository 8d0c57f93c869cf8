#!/bin/bash
# GPU suite on the checked build (libmarlin_gpu_checked.so, make gpu-checked), then -- only if that was clean -- on the
# product build, then the bench: tools/checked_then_plain.sh <tag>
set -e -o pipefail
out=$PWD/gpurun_out/${1:-ctp}
mkdir -p $out
L=marlin_amd/lib
cp $L/libmarlin_gpu.so $out/plain.so
cp $L/libmarlin_gpu_checked.so $L/libmarlin_gpu.so
timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests_checked.log 2>&1 || { tail -20 $out/tests_checked.log; cp $out/plain.so $L/libmarlin_gpu.so; rm -f $out/plain.so; exit 1; }
tail -2 $out/tests_checked.log
timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 3 --warmup 1 > $out/bench_checked.json 2> $out/bench_checked.err || { tail -5 $out/bench_checked.err; cp $out/plain.so $L/libmarlin_gpu.so; rm -f $out/plain.so; exit 1; }
cp $out/plain.so $L/libmarlin_gpu.so; rm -f $out/plain.so
timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests_plain.log 2>&1 || { tail -20 $out/tests_plain.log; exit 1; }
tail -2 $out/tests_plain.log
timeout -k 10 200 python3 bench.py --no-cpu --no-latency --steps 10 --warmup 3 > $out/bench_plain.json 2> $out/bench_plain.err
python3 -c "
import json;d=json.load(open('$out/bench_plain.json'));print(d['value']);print(json.dumps(d['stages']))"
