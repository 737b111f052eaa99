set -e
cd /root/repo
for v in exp/qt64.so exp/qt128.so; do
 for a in "" "--kind noise" "--workload C1_tum"; do
  n=$(echo "$v$a" | tr -d ' -/.' )
  ORBHIP_LIB=${v:-visual_slam_amd/libORBhip.so} timeout -k 10 300 python bench.py --cpu-frames 0 $a > gpurun_out/s3_$n.json 2> gpurun_out/s3_$n.err
  python - gpurun_out/s3_$n.json "$v $a" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], d['roofline']['kernel_ms_per_step'])
PY
 done
done
