#!/bin/bash
# rocprofv3 capture of ONE workload's render kernel, serial launches with the library's default options
# (bench.py --frames-in-flight 1): kernel trace + stats, then PMC passes, each alone and never with sys / hip traces
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  The program follows `--` directly.
# usage: scripts/profile_round.sh <workload> [bench args...]   -> gpurun_out/prof_<round>_<workload>/, digested by
#        scripts/make_pmc_json.py into profiles/<round>/pmc_<workload>.json + pmc_<workload>.txt   (RM_ROUND, default r03)
set -e
WL=${1:-C3}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${RM_ROUND:-r03}
export RM_ROUND=$RND
OUT=$R/gpurun_out/prof_${RND}_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-verify $@"
echo "$ARGS" > $OUT/bench_args.txt
run() {  # name, rocprofv3 args...
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass failed: $name"
  echo "pass $name done"
}
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run pmc_mix1 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT
run pmc_mix2 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32
run pmc_mix3 --pmc SQ_INSTS_VALU_INT64 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
python3 $R/scripts/make_pmc_json.py $WL $OUT
