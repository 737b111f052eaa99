#!/bin/bash
# Collects the profiles of one round on the GPU box: kernel trace of the default bench command and
# the two HBM traffic counter passes (each in a run of its own). Output under gpurun_out/$1/.
set -e
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out -o trace -- python3 bench.py --cpu-frames 0 > $out/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out -o fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out -o write -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 > $out/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out -o sq -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 > $out/sq.log 2>&1
ls $out
