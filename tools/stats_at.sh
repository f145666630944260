# kernel stats of a bench configuration (development aid): bash tools/stats_at.sh "<bench args>" <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stats_$2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-extras $1 > $O/log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$O/st/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
