#!/bin/bash
# level-synchronous active-set pipeline: tests, then timings per pass count and kick scale
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3as; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_as_pipeline.py -x -q 2>&1 | tail -15 | tee $O/pytest_new.log
cd /tmp; export TMPDIR=/tmp
for ks in 1 2 3; do for ap in 0 1 2 3 4; do
export CFNMPC_AS_PASSES=$ap
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --kick-scale $ks --steps 20 --warmup 20 --no-cpu-baseline --no-extras > $O/ks_${ks}_$ap.log 2>&1
grep "^{" $O/ks_${ks}_$ap.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('== kick $ks as_passes $ap:', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],3), 'ms', d['qp_stats'])"
python - <<PY
import csv,glob
f=glob.glob('/tmp/ks/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if any(k in r['Name'] for k in ('k_as','k_ipm','k_asp','k_ascommit')):
        print('   ', r['Name'][:44].ljust(44), r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done; done
unset CFNMPC_AS_PASSES
cd $R
CFNMPC_AS_PASSES=3 timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_all_ap3.log
