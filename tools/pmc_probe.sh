#!/bin/bash
# GPU box: three rocprofv3 --pmc passes over one command (SQ issue/wait split + MFMA busy + LDS conflicts; L2 requests / hits;
# HBM fetch), CSV output under gpurun_out/$1/.  Usage: tools/pmc_probe.sh <tag> <python script> [args...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/"$@" > $O/sq.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/tcc -- python3 $R/"$@" > $O/tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/"$@" > $O/fetch.log 2>&1
cd $R && python3 tools/pmc_csv.py $O
