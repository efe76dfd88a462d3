# PMC passes (separate runs, kernel-trace only) of a short bench run.  Usage: bash tools/pmc_quick.sh <tag> [bench args]
tag=${1:-q}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_pmc_$n -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-preheat "$@" > gpurun_out/${tag}_pmc_$n.log 2>&1
  python tools/rocprof_summary.py gpurun_out/${tag}_pmc_$n/p_results.db --pmc 2>&1 | awk '/## counters/{f=1} f' > gpurun_out/${tag}_pmc_$n.md
  rm -rf gpurun_out/${tag}_pmc_$n
done
cat gpurun_out/${tag}_pmc_*.md | grep -v "^$" | cut -c1-400
