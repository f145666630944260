# A/B of library builds on config C5 (mixed horizons, one GPU): bench.py --workload mixed and the plain-loop side figure
# usage: LIBS="libcfnmpc_b.so libcfnmpc.so" bash tools/ab_mixed.sh [reps]
cd $GRAFT_REPO_ROOT
REPS=${1:-3}
LIBS=${LIBS:-libcfnmpc.so}
for i in $(seq $REPS); do for lib in $LIBS; do
  CFNMPC_LIB=$GRAFT_REPO_ROOT/crazyflie_nmpc_amd/$lib python bench.py --workload mixed --steps 40 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'C5 delay loop', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms', round(d['stage_steps_per_s']/1e6,1), 'M stage-steps/s')"
  CFNMPC_LIB=$GRAFT_REPO_ROOT/crazyflie_nmpc_amd/$lib python tools/mixed_bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'C5 plain loop', round(d['rti_steps_per_s']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms', round(d['stage_steps_per_s']/1e6,1), 'M stage-steps/s')"
done; done
