# A/B on one box at the small fleet sizes (development aid): libraries x step_graph, no events in the timed steps
# usage: LIBS="libcfnmpc_b.so libcfnmpc.so" SIZES="4096 8192 16384" GRAPH="0 1" bash tools/ab_small.sh [reps]
cd $GRAFT_REPO_ROOT
REPS=${1:-3}
LIBS=${LIBS:-libcfnmpc.so}
SIZES=${SIZES:-4096 8192 16384}
GRAPH=${GRAPH:-0}
for B in $SIZES; do for i in $(seq $REPS); do for lib in $LIBS; do for g in $GRAPH; do
  CFNMPC_LIB=$GRAFT_REPO_ROOT/crazyflie_nmpc_amd/$lib python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-extras --no-profile --batch $B --step-graph $g $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'B=$B', 'graph=$g', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms', 'constrained', round(d['qp_stats']['frac_constrained'],4), 'solves', round(d['qp_stats']['mean_qp_solves'],4), 'head', round(d['qp_stats']['mean_head_stages'],3))"
done; done; done; done
