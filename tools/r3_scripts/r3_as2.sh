#!/bin/bash
# AS scheduling variants by fleet size (internal encoding: 0 mono, -1 all-in-one + commit, p passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3as2; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_as_pipeline.py -x -q 2>&1 | tail -5 | tee $O/pytest_new.log
cd /tmp; export TMPDIR=/tmp
for bs in 4096 8192 65536; do for ap in 0 -1 1 2 3; do
export CFNMPC_AS_PASSES=$ap
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --batch $bs --steps 40 --warmup 40 --no-cpu-baseline --no-extras > $O/b_${bs}_$ap.log 2>&1
grep "^{" $O/b_${bs}_$ap.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('== batch $bs as_passes $ap:', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4))"
python - <<PY
import csv,glob
f=glob.glob('/tmp/ks/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if any(k in r['Name'] for k in ('k_as','k_ipm','k_asp','k_ascommit','k_factor','k_forward','k_linearise','k_compact','k_scatter','k_rank')):
        print('   ', r['Name'][:44].ljust(44), r['Calls'], 'avg', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done; done
