#!/bin/bash
# GPU box: counters of k_nuts<DenseMvnCoop> at configs[3], single-transition launches and one launch of 20 transitions
# (tools/ubench/dense_nuts_pmc_run.py).  One rocprofv3 run per counter group; outputs under gpurun_out/dense_nuts_pmc/<group>/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dense_nuts_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/ubench/dense_nuts_pmc_run.py > $OUT/trace.log 2>&1
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "busy SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" "grbm GRBM_GUI_ACTIVE GRBM_COUNT"; do
    set -- $grp; g=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/$g -- python3 $R/tools/ubench/dense_nuts_pmc_run.py > $OUT/$g.log 2>&1
    echo "$g done"
done
cd $R && python3 tools/summarize_dense_nuts_pmc.py
