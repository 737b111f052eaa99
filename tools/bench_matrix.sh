#!/bin/bash
# Bench lines for the three synthetic distributions of the headline workload, the input variants, the
# 4K workload, the host-fed (PCIe-inclusive) rate, the dataset-sized workloads and the stereo chain.
# Output: gpurun_out/matrix/*.json
mkdir -p gpurun_out/matrix
run() { name=$1; shift; timeout -k 10 300 python bench.py --cpu-frames 0 "$@" > gpurun_out/matrix/$name.json 2> gpurun_out/matrix/$name.err || exit 1; }
for kind in blocks noise lowc; do run c4_$kind --kind $kind; done
run c4_bgr --color 3
run c4_rectify --rectify
run c5_blocks --workload C5_4k_8000f --batch 128
run c4_hostfed --host-fed
for w in C1_tum C2_kitti C3_euroc; do run $w --workload $w --batch 1024; done
run c2_stereo --workload C2_kitti --stereo --batch 256
run c4_stereo --stereo --batch 256
