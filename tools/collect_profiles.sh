#!/bin/bash
# Copies what tools/pmc_round.sh left under gpurun_out/$1 into profiles/ under the names of round $2 (e.g. r03):
# kernel traces, counter passes, per-kernel reports and the traffic / instruction summaries bench.py reads.
src=gpurun_out/${1:?tag}; r=${2:?round}
( cd $src && python3 ../../tools/pmc_report.py . ext 512 ./$r > ext_report.json ) || exit 1
cp $src/ext_trace.txt profiles/${r}_ext_kernel_stats.txt
cp $src/ext_trace_alone.txt profiles/${r}_ext_kernel_stats_one_at_a_time.txt
for p in st m; do [ -f $src/${p}_trace.txt ] && cp $src/${p}_trace.txt profiles/${r}_${p}_kernel_stats.txt; done
for p in ext st m; do
  for c in lds issue fetch write clk; do [ -f $src/${p}_$c.txt ] && cp $src/${p}_$c.txt profiles/${r}_pmc_${p}_$c.txt; done
  [ -f $src/${p}_report.json ] && cp $src/${p}_report.json profiles/${r}_pmc_${p}_report.json
done
sed "s#(\./ext_#(gpurun_out/$1/ext_#" $src/${r}_traffic.json > profiles/${r}_traffic.json
sed "s#(\./ext_#(gpurun_out/$1/ext_#" $src/${r}_valu.json > profiles/${r}_valu.json
git status --short profiles | head -40
