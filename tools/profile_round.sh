#!/bin/bash
# Round snapshot on the GPU box (run through gpurun): kernel stats, the two PMC passes, the bench
# line.  Everything lands in gpurun_out/snap/; copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/snap
rm -rf $O; mkdir -p $O   # (the local gpurun_out/ MERGES files of successive calls: clear gpurun_out/snap before pulling a new snapshot)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 6 --warmup 20 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 6 --warmup 20 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_summary.csv
cp $O/pmc_traffic.json $R/profiles/pmc_traffic_latest.json   # so that the bench line below carries the fresh figure
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
cd $R && python bench.py 2>/dev/null | tail -1 > $O/bench_line.json
cat $O/bench_line.json
