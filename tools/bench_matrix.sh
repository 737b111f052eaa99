#!/bin/bash
# Bench lines for the three synthetic distributions of the headline workload, the 4K workload and
# the host-fed (PCIe-inclusive) rate. Output: gpurun_out/matrix/*.json
mkdir -p gpurun_out/matrix
for kind in blocks noise lowc; do
  timeout -k 10 300 python bench.py --cpu-frames 0 --kind $kind > gpurun_out/matrix/c4_$kind.json 2> gpurun_out/matrix/c4_$kind.err || exit 1
done
timeout -k 10 300 python bench.py --cpu-frames 0 --workload C5_4k_8000f --batch 32 > gpurun_out/matrix/c5_blocks.json 2> gpurun_out/matrix/c5.err || exit 1
timeout -k 10 300 python bench.py --cpu-frames 0 --host-fed > gpurun_out/matrix/c4_hostfed.json 2> gpurun_out/matrix/hostfed.err || exit 1
for w in C1_tum C2_kitti C3_euroc; do
  timeout -k 10 300 python bench.py --cpu-frames 0 --workload $w --batch 512 > gpurun_out/matrix/$w.json 2> gpurun_out/matrix/$w.err || exit 1
done
