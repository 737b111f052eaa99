set -e
cd /root/repo
for a in "--kind noise" "--workload C5_4k_8000f --batch 128" "--workload C1_tum" "--workload C2_kitti" "--workload C5_4k_8000f --batch 128 --kind noise"; do
  n=$(echo $a | tr -d ' -' )
  timeout -k 10 300 python bench.py --cpu-frames 0 $a > gpurun_out/s2_$n.json 2> gpurun_out/s2_$n.err
done
