#!/bin/bash
# the round's measurement passes on the GPU box: tools/gpu_profile.sh <tag>  -> gpurun_out/<tag>/
# every GPU step under its own time limit; the first failure ends the script
set -e -o pipefail
tag=${1:-prof}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench_plain.json 2> $out/bench_plain.err
tail -c 3000 $out/bench_plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_if1 -o kt -- python3 bench.py --no-cpu --no-latency --inflight 1 --steps 1 --warmup 1 --sub 32 > $out/bench_if1.json 2> $out/kt_if1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_def -o kt -- python3 bench.py --no-cpu --no-latency > $out/bench_def_profiled.json 2> $out/kt_def.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 bench.py --no-cpu --no-latency --inflight 1 --steps 1 --warmup 0 --sub 3 > $out/pmc_fetch.json 2> $out/pmc_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 bench.py --no-cpu --no-latency --inflight 1 --steps 1 --warmup 0 --sub 3 > $out/pmc_write.json 2> $out/pmc_write.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq -o pmc -- python3 bench.py --no-cpu --no-latency --inflight 1 --steps 1 --warmup 0 --sub 3 > $out/pmc_sq.json 2> $out/pmc_sq.err
find $out -name '*kernel_trace.csv' -delete
find $out -name '*.csv' | head -n 20 || true
