cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA_RDREQ_sum SQ_WAVES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/q_$n -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/q_$n.log 2>&1
  python tools/rocprof_summary.py gpurun_out/q_$n/p_results.db --pmc 2>&1 | awk '/## counters/{f=1} f' | grep -E "kernel \||conv3|stem_dd|conv_mfma_kernel<1, 2, 0" | cut -c1-200
  python tools/rocprof_summary.py gpurun_out/q_$n/p_results.db 2>&1 | grep -E "conv3|stem_dd" | cut -c1-110
done
rm -rf gpurun_out/q_*
