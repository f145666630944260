#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3as3; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_as_pipeline.py -x -q 2>&1 | tail -5 | tee $O/pytest_new.log
cd /tmp; export TMPDIR=/tmp
for bs in 2048 4096 8192 16384 65536; do for ap in 0 -2; do
export CFNMPC_AS_PASSES=$ap
for rep in 1 2; do
python $R/bench.py --batch $bs --steps 60 --warmup 40 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('== batch $bs as_passes $ap:', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms kernels', {k[:12]: round(v,4) for k,v in d['roofline']['kernels_ms'].items()})"
done; done; done
