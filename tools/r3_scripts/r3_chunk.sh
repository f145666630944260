#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3chunk; rm -rf $O; mkdir -p $O
cd $R
python tools/chunked_pair.py check 2>&1 | tail -3
python tools/chunked_pair.py 65536 2>&1 | grep "^batch" | tee $O/timing.log
cd /tmp; export TMPDIR=/tmp
for c in 0 1 2 5; do
for ctr in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pp -- python $R/tools/chunked_pair.py 65536 $c > /dev/null 2>&1
echo "chunk $c: $(python $R/tools/pmc_pair.py /tmp/pp $ctr 39 | tr '\n' ';')" | tee -a $O/pmc.log
done; done
