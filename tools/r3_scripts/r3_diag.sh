#!/bin/bash
# round-3 baseline diagnostics: QP statistics + kernel stats per kick scale
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3diag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/tools/qp_diag.py 1 2 3 > $O/qp_diag.log 2>&1
cat $O/qp_diag.log
for ks in 1 2 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$ks -- python $R/bench.py --kick-scale $ks --steps 20 --warmup 20 --no-cpu-baseline --no-extras > $O/ks_$ks.log 2>&1
tail -1 $O/ks_$ks.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kick $ks', round(d['value']/1e6,3), round(d['ms_per_step'],3), d['qp_stats'])"
python - <<PY
import csv,glob
f=glob.glob('$O/ks_$ks/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r['Name'][:40].ljust(40), r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done
