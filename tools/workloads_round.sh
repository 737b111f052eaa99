#!/bin/bash
# Bench lines of every workload, preset, input kind and batch size of a round: gpurun_out/$1/workloads.jsonl (one JSON
# line per run, with a "run" key naming it), the default bench line (with the CPU baseline) in bench.json, single-frame
# latencies in latency.txt, the matcher and front-end benches beside them.
tag=${1:-workloads}; out=gpurun_out/$tag; mkdir -p $out; : > $out/workloads.jsonl
run() { name=$1; shift
  timeout -k 10 300 python bench.py --cpu-frames 0 "$@" > $out/line.json 2> $out/line.err || { echo "$name failed"; tail -3 $out/line.err; return 1; }
  python3 - "$out/line.json" "$name" >> $out/workloads.jsonl <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); d={"run": sys.argv[2], **d}; print(json.dumps(d))
PY
  python3 - "$out/line.json" "$name" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%-26s %10.0f %-8s frac %.4f  %s" % (sys.argv[2], d["value"], d["unit"], r["whole_path_frac"], r["kernel_ms_per_step"]))
PY
}
timeout -k 10 300 python bench.py --host-fed > $out/bench.json 2> $out/bench.err || { tail -3 $out/bench.err; exit 1; }
run c4_blocks && run c4_noise --kind noise --batch 256 && run c4_lowc --kind lowc &&
run c1_tum --workload C1_tum && run c2_kitti --workload C2_kitti && run c3_euroc --workload C3_euroc && run c5_4k --workload C5_4k_8000f --batch 128 &&
run preset_c4_strong_1024 --preset c4_strong_1024 && run preset_c3_stereo_512 --preset c3_stereo_512 &&
run per_rank_c4_128_of_1024 --preset c4_strong_1024 --total-batch 128 --steps 30 && run per_rank_c3_stereo_64_of_512 --preset c3_stereo_512 --total-batch 64 --steps 30 &&
run c4_bgr --color 3 && run c4_rectify --rectify && run c2_stereo --workload C2_kitti --stereo --batch 256 && run c4_stereo --stereo --batch 256 &&
run c4_b64 --batch 64 --steps 40 && run c4_b128 --batch 128 --steps 30 && run c4_b256 --batch 256 --steps 20 && run c4_b512 --batch 512 --steps 10 || exit 1
rm -f $out/line.json $out/line.err
: > $out/latency.txt
for w in C1_tum C3_euroc C2_kitti C4_1080p_2000f; do timeout -k 10 60 python tools/latency.py $w >> $out/latency.txt 2>&1; done
cat $out/latency.txt
timeout -k 10 300 python tools/match_bench.py > $out/match_bench.jsonl 2> $out/match_bench.err; tail -2 $out/match_bench.err
timeout -k 10 300 python tools/frontend_bench.py > $out/frontend_bench.jsonl 2> $out/frontend_bench.err; tail -2 $out/frontend_bench.err
