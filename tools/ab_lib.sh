# A/B of several builds of the library on one box (development aid): CFNMPC_LIB selects the .so
# usage: LIBS="libcfnmpc_prev.so libcfnmpc.so" bash tools/ab_lib.sh "<bench args>" [reps]
cd $GRAFT_REPO_ROOT
ARGS=${1:---batch 4096}
REPS=${2:-2}
LIBS=${LIBS:-libcfnmpc_prev.so libcfnmpc.so}
for i in $(seq $REPS); do
for lib in $LIBS; do
  CFNMPC_LIB=$GRAFT_REPO_ROOT/crazyflie_nmpc_amd/$lib python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-extras $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', '$ARGS', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms', {k[:12]: round(v,4) for k,v in (r.get('kernels_ms') or {}).items()})"
done; done
