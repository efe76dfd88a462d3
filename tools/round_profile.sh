# One-shot evidence run on the GPU box: full GPU tests, bench, rocprofv3 kernel trace + PMC passes (separate runs).
# Usage: bash tools/round_profile.sh <tag>     (outputs under gpurun_out/<tag>_*)
tag=${1:-r1}
python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_trace.log 2>&1
python tools/rocprof_summary.py gpurun_out/${tag}_trace/p_results.db > gpurun_out/${tag}_kernel_trace.md 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_pmc_$n -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_$n.log 2>&1
  python tools/rocprof_summary.py gpurun_out/${tag}_pmc_$n/p_results.db --pmc 2>&1 | awk '/## counters/{f=1} f' > gpurun_out/${tag}_pmc_$n.md
done
head -12 gpurun_out/${tag}_kernel_trace.md
