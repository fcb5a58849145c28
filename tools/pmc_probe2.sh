#!/bin/bash
# GPU box: one rocprofv3 --pmc pass with the issue-side SQ counters (VALU / LDS / MFMA / waiting) over one command.
# Usage: tools/pmc_probe2.sh <tag> <python script> [args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/"$@" > $O/sq.log 2>&1
cd $R && python3 tools/pmc_csv.py $O
