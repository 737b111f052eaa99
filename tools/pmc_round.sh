#!/bin/bash
# Counter passes of one round on the GPU box (each in a run of its own, program directly after `--`):
# LDS activity / bank conflicts, issue and stall split, HBM fetch / write sizes and the GPU busy clock,
# for the extraction bench (k_level0, k_resize, k_fast, k_quadtree, k_describe), the stereo bench
# (k_stereo_*) and the matcher bench (k_search_*). Output under gpurun_out/$1/.
tag=${1:-pmc}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
ISSUE="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
run() {   # name, counters, program...: the per-kernel sums go to $out/name.txt, the database is deleted (it is large)
    local name=$1 ctr=$2; shift 2
    timeout -s KILL 150 rocprofv3 --pmc $ctr -d $out -o $name -- "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -15 $out/$name.log; return 1; }
    local db=$(find $out -name "${name}_results.db" | head -1)
    python3 tools/pmc_summary.py "$db" > $out/$name.txt 2>&1 && rm -f "$db" && tail -4 $out/$name.log > $out/$name.log.tail && rm -f $out/$name.log
}
trace() {   # name, program...: kernel durations (rocprofv3 --kernel-trace --stats) as text
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o $name -- "$@" > $out/$name.log 2>&1 || { echo "trace $name failed"; tail -15 $out/$name.log; return 1; }
    local db=$(find $out -name "${name}_results.db" | head -1)
    python3 tools/prof_summary.py "$db" > $out/$name.txt 2>&1 && rm -f "$db" $out/$name.log
    find $out -name "${name}_*.csv" -delete
}
B="python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --batch 512 --skip-alone"
# kernel durations twice: as the bench runs them (level 0's FAST beside the resize chain: each of the two takes longer
# than alone, the pair less than one after the other) and one kernel at a time (ORBHIP_OVERLAP=0), which is also how
# the counter passes see them (the profiler serialises kernels while it collects counters)
trace ext_trace python3 bench.py --cpu-frames 0 --batch 512 --skip-alone && ORBHIP_OVERLAP=0 trace ext_trace_alone python3 bench.py --cpu-frames 0 --batch 512 --skip-alone && run ext_lds "$LDS" $B && run ext_issue "$ISSUE" $B && run ext_fetch FETCH_SIZE $B && run ext_write WRITE_SIZE $B && run ext_clk GRBM_GUI_ACTIVE $B || exit 1
if [ "$2" != "ext" ]; then
    S="python3 bench.py --stereo --workload C2_kitti --steps 2 --warmup 1 --cpu-frames 0 --batch 256"
    trace st_trace $S && run st_lds "$LDS" $S && run st_issue "$ISSUE" $S && run st_fetch FETCH_SIZE $S && run st_write WRITE_SIZE $S && run st_clk GRBM_GUI_ACTIVE $S || exit 1
    M="python3 tools/match_bench.py --steps 2 --cpu-pairs 0"
    trace m_trace $M && run m_lds "$LDS" $M && run m_issue "$ISSUE" $M && run m_fetch FETCH_SIZE $M && run m_write WRITE_SIZE $M && run m_clk GRBM_GUI_ACTIVE $M || exit 1
fi
for p in ext st m; do [ -f $out/${p}_lds.txt ] && python3 tools/pmc_report.py $out $p > $out/${p}_report.json; done
ls $out | head -60
