#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
make -C tests/harness -s
python - <<'PY'
import numpy as np, subprocess, os
x0=np.array([0.25,-0.2,0.55,1,0,0,0,0.1,-0.1,0.05,0,0,0.0]); np.savetxt('/tmp/x0.txt', x0[None])
for ap in ('0','-2'):
    env=dict(os.environ, CFNMPC_AS_PASSES=ap)
    r=subprocess.run(['tests/harness/cf_nmpc_replay','regulation','-','300','/tmp/x0.txt','1','/tmp/out.csv'],env=env,capture_output=True,text=True)
    print('B=1 as_passes',ap,r.stderr.strip().splitlines()[-1])
PY
for bs in 4096 8192; do for rep in 1 2; do
python bench.py --batch $bs --steps 60 --warmup 40 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('== batch $bs:', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4))"
done; done
