export TMPDIR=/tmp
out=gpurun_out/pmc_noise; mkdir -p $out
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
ISSUE="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
B="python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 512 --kind noise"
for p in lds issue clk; do
  case $p in lds) C="$LDS";; issue) C="$ISSUE";; clk) C="GRBM_GUI_ACTIVE";; esac
  timeout -s KILL 150 rocprofv3 --pmc $C -d $out -o ext_$p -- $B > $out/ext_$p.log 2>&1 || { echo fail $p; tail -5 $out/ext_$p.log; exit 1; }
  db=$(find $out -name "ext_${p}_results.db" | head -1)
  python3 tools/pmc_summary.py "$db" > $out/ext_$p.txt 2>&1 && rm -f "$db" $out/ext_$p.log
done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o ext_trace -- python3 bench.py --cpu-frames 0 --batch 512 --kind noise > $out/ext_trace.log 2>&1
db=$(find $out -name "ext_trace_results.db" | head -1); python3 tools/prof_summary.py "$db" > $out/ext_trace.txt 2>&1 && rm -f "$db" $out/ext_trace.log; find $out -name "*.csv" -delete
python3 tools/pmc_report.py $out ext > $out/ext_report.json 2>&1; cat $out/ext_report.json | head -40
