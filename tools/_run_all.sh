set -e
cd /root/repo
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/all_tests.log 2>&1 || { tail -30 gpurun_out/all_tests.log; exit 1; }
tail -3 gpurun_out/all_tests.log
timeout -k 10 300 python bench.py --cpu-frames 0 > gpurun_out/all_bench.json 2> gpurun_out/all_bench.err
python - gpurun_out/all_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel_ms_per_step'])
PY
