set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_inplace.py tests/test_gpu_asbuilt.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4
for a in "" "--workload C1_tum"; do
  timeout -k 10 300 python bench.py --cpu-frames 0 $a > gpurun_out/s4.json 2> gpurun_out/s4.err
  python - gpurun_out/s4.json "$a" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], d['roofline']['kernel_ms_per_step'])
PY
done
