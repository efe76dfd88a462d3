# PMC passes over the stem kernel (rocprofv3; counters only together with --kernel-trace).  Usage: bash tools/stem_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-stem}
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_IFETCH"; do
  n=$(echo $set | cut -d" " -f1)
  PNVO_STEM_DBG=${DBG:-0} rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_$n -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_$n.log 2>&1
  python tools/rocprof_summary.py gpurun_out/${tag}_$n/p_results.db --pmc 2>&1 | grep -E "^\| kernel \| grid \| [A-Z]|stem_" | grep -v calls | head -4
done
