set -e
cd /root/repo
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/s2_tests.log 2>&1
timeout -k 10 200 python bench.py --cpu-frames 0 > gpurun_out/s2_bench_default.json 2> gpurun_out/s2_bench_default.err
timeout -k 10 200 python bench.py --cpu-frames 0 --fast-th 20,20 > gpurun_out/s2_bench_th2020.json 2> gpurun_out/s2_bench_th2020.err
timeout -k 10 200 python bench.py --cpu-frames 0 --kind lowc > gpurun_out/s2_bench_lowc.json 2> gpurun_out/s2_bench_lowc.err
timeout -k 10 200 python bench.py --cpu-frames 0 --kind lowc --fast-th 20,20 > gpurun_out/s2_bench_lowc2020.json 2> gpurun_out/s2_bench_lowc2020.err
tail -3 gpurun_out/s2_tests.log
