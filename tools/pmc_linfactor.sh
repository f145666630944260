#!/bin/bash
# SQ counters of the start solve's backward half (development aid; run through gpurun): tools/linfactor_time.py under one --pmc pass
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/lf_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/tools/linfactor_time.py 65536 > $O/sq.log 2>&1
python $R/tools/pmc_sq.py $O/sq k_linfactor k_factor k_linearise | tee $O/sq_summary.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAVES --output-format csv -d $O/sq2 -- python $R/tools/linfactor_time.py 65536 > $O/sq2.log 2>&1
python $R/tools/pmc_sq.py $O/sq2 k_linfactor k_factor k_linearise | tee -a $O/sq_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/linfactor_time.py 65536 > $O/stats.log 2>&1
cat $O/stats/*/*kernel_stats.csv | head -6
tail -3 $O/stats.log
