# active-horizon parameters against the tail-check retries of k_as (development aid)
cd $GRAFT_REPO_ROOT
for ex in ${EXS:-4 6 8}; do for mg in ${MGS:-0.10 0.15 0.20}; do
python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-extras --ah-extra $ex --ah-margin $mg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ah_extra $ex ah_margin $mg', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms k_as', round(r['kernels_ms']['k_as (active-set solves, roll-out)'],4), 'mean head', round(d['qp_stats']['mean_head_stages'],3))"
done; done
