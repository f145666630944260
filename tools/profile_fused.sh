#!/bin/bash
# The fused start solve (bench.py --start-solve 2) under the round snapshot's three passes: kernel stats, FETCH_SIZE, WRITE_SIZE
# (separate --pmc runs, kernel trace only).  Results in gpurun_out/snap_fused/; copied to profiles/r04_fused_* by hand.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/snap_fused
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--start-solve 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $A > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 6 --warmup 20 $A > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 6 --warmup 20 $A > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_summary.csv
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
head -9 $O/kernel_stats.csv
cd $R && python bench.py $A 2>/dev/null | tail -1 > $O/bench_line.json
python -c "import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
