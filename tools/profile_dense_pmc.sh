#!/bin/bash
# GPU box: counter evidence for the dense single-step sweep (configs[3]) with lanes (default) and without (IDHMC_DENSE_LANES=0).
# One rocprofv3 run per counter group; outputs under gpurun_out/dense_pmc/<tag>/<group>/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dense_pmc
rm -rf $OUT; mkdir -p $OUT
for tag in lanes4 lanes0; do
    if [ $tag = lanes0 ]; then export IDHMC_DENSE_LANES=0; else export IDHMC_DENSE_LANES=4; fi
    SWEEPS=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -- python3 $R/tools/ubench/dense_pmc_run.py > $OUT/$tag.trace.log 2>&1
    for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
               "tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "grbm GRBM_GUI_ACTIVE GRBM_COUNT"; do
        set -- $grp; g=$1; shift
        SWEEPS=200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag/$g -- python3 $R/tools/ubench/dense_pmc_run.py > $OUT/$tag.$g.log 2>&1
        echo "$tag $g done"
    done
done
