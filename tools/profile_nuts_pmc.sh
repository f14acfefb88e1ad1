#!/bin/bash
# GPU box: counter evidence for k_nuts (configs[2]'s kernel) at tree depth 4 (eps 0.25) and depth 7 (eps 0.03).
# One rocprofv3 run per counter group (the guide: FETCH_SIZE and WRITE_SIZE cannot share a pass; --pmc never together
# with a trace domain other than --kernel-trace).  Outputs under gpurun_out/nuts_pmc/<tag>/<group>/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/nuts_pmc
mkdir -p $OUT
export NT=${NT:-3} C=${C:-65536} D=${D:-1024}
run_group() {   # tag eps group counters...
    local tag=$1 eps=$2 group=$3; shift 3
    EPS=$eps rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag/$group -- python3 $R/tools/bench_nuts.py > $OUT/$tag.$group.log 2>&1
    echo "$tag $group done: $(grep 'steps/s' $OUT/$tag.$group.log | cut -c1-120)"
}
for cfg in "d4 0.25" "d7 0.03"; do
    set -- $cfg; tag=$1; eps=$2
    EPS=$eps rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -- python3 $R/tools/bench_nuts.py > $OUT/$tag.trace.log 2>&1
    echo "$tag trace done: $(grep 'steps/s' $OUT/$tag.trace.log | cut -c1-120)"
    run_group $tag $eps fetch FETCH_SIZE
    run_group $tag $eps write WRITE_SIZE
    run_group $tag $eps sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
    run_group $tag $eps sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES
    run_group $tag $eps tcc TCC_HIT_sum TCC_MISS_sum
    run_group $tag $eps grbm GRBM_GUI_ACTIVE GRBM_COUNT
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
