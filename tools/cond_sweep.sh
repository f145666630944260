#!/bin/bash
# N2 sweep of the partial-condensing option on the GPU box: per N2 the bench
# line, rocprofv3 kernel stats and the two PMC passes (HBM bytes per kernel).  Output: gpurun_out/cond/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cond
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N2 in ${1:-0 25 10 5}; do
  A="--no-cpu-baseline --no-extras --steps 10 --warmup 10"
  if [ "$N2" != "0" ]; then A="$A --cond-n2 $N2"; else A="$A --active-set 0 --active-horizon 0"; fi
  python $R/bench.py $A 2>/dev/null | tail -1 > $O/bench_n2_$N2.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$N2 -- python $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$N2 -- python $R/bench.py $A --steps 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$N2 -- python $R/bench.py $A --steps 4 > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $O/f_$N2 $O/w_$N2 $O/traffic_$N2.json $O/pmc_$N2.csv > /dev/null
  cp $O/stats_$N2/*/*kernel_stats.csv $O/kernel_stats_$N2.csv
  rm -rf $O/stats_$N2 $O/f_$N2 $O/w_$N2
  echo "== N2 = $N2"; python -c "import json; d=json.load(open('$O/bench_n2_$N2.json')); print(d['value'], d['ms_per_step'], d['roofline']['linearise_ms'], d['roofline']['qp_ms'], d['qp_stats'])"
  head -7 $O/kernel_stats_$N2.csv | cut -d, -f1-4
  cat $O/pmc_$N2.csv | grep -v "^cfn::k_\(get\|put\|sim\|init\)"
done
