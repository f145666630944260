cd $GRAFT_REPO_ROOT
for m in 1 2 1 2; do python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --start-solve $m 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('start_solve $m', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],3), 'ms kernels', r.get('kernels_ms'))"; done
