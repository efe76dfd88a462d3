# One-shot evidence run on the GPU box: benches of the three configurations, rocprofv3 kernel traces and PMC passes
# (separate runs, kernel-trace only), profiles/traffic.json.  Usage: bash tools/round_profile.sh <tag>
# Outputs land in gpurun_out/<tag>_* ; copy what should be judged into profiles/.
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for cfg in fwd_fp32 dual_bf16; do
  rocprofv3 --kernel-trace --stats -d $O/${tag}_trace_$cfg -o p -- python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/${tag}_trace_$cfg.log 2>&1
  python tools/rocprof_summary.py $O/${tag}_trace_$cfg/p_results.db > $O/${tag}_kernel_trace_$cfg.md 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    n=$(echo $set | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $set -d $O/${tag}_pmc_${cfg}_$n -o p -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-preheat --no-secondary > $O/${tag}_pmc_${cfg}_$n.log 2>&1
    python tools/rocprof_summary.py $O/${tag}_pmc_${cfg}_$n/p_results.db --pmc 2>&1 | awk '/## counters/{f=1} f' > $O/${tag}_pmc_${cfg}_$n.md
  done
  python tools/make_traffic.py $cfg 256 $O/${tag}_pmc_${cfg}_FETCH_SIZE/p_results.db $O/${tag}_pmc_${cfg}_WRITE_SIZE/p_results.db $O/${tag}_traffic.json $O/${tag}_kernel_trace_$cfg.md
  cat $O/${tag}_pmc_${cfg}_*.md > $O/${tag}_pmc_$cfg.md
  python tools/mfma_util.py $O/${tag}_kernel_trace_$cfg.md $O/${tag}_pmc_$cfg.md > $O/${tag}_mfma_util_$cfg.md
  rm -rf $O/${tag}_pmc_${cfg}_* $O/${tag}_trace_$cfg
done
# the three bench lines AFTER the counter passes, with this run's traffic record in place (bench.py reads profiles/traffic.json and
# accepts it only for the kernel sources it was measured on)
cp $O/${tag}_traffic.json profiles/traffic.json
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err; tail -c 300 $O/${tag}_bench.json; echo
python bench.py --config dual_bf16 > $O/${tag}_bench_dual_bf16.json 2> $O/${tag}_bench_dual_bf16.err; tail -c 300 $O/${tag}_bench_dual_bf16.json; echo
python bench.py --config train --steps 10 --warmup 3 > $O/${tag}_bench_train.json 2> $O/${tag}_bench_train.err; tail -c 300 $O/${tag}_bench_train.json; echo
# training step: kernel trace only (its kernels are MFMA / HBM bound by construction; counters are taken for the forward)
rocprofv3 --kernel-trace --stats -d $O/${tag}_trace_train -o p -- python bench.py --config train --steps 10 --warmup 3 > $O/${tag}_trace_train.log 2>&1
python tools/rocprof_summary.py $O/${tag}_trace_train/p_results.db > $O/${tag}_kernel_trace_train.md 2>&1
rm -rf $O/${tag}_trace_train
PNVO_WSM_PROF=1 python bench.py --config train --steps 10 --warmup 3 2>&1 >/dev/null | grep "pnvo\]" > $O/${tag}_wgrad_stem_phases.txt
PNVO_X3_PROF=1 python bench.py --steps 1 --warmup 0 --no-preheat --no-cpu-baseline --no-secondary 2>&1 >/dev/null | grep "pnvo\] conv_x3" | sort -u -t: -k1,1 > $O/${tag}_conv_x3_phases.txt
PNVO_STEM_FORM=tiles PNVO_STEM_DBG=9 python bench.py --steps 3 --warmup 1 --no-preheat --no-cpu-baseline --no-secondary 2>&1 >/dev/null | grep "pnvo\] stem_mx" > $O/${tag}_stem_phases.txt
# the resident-weight stem (the default): cycles per tile of its three sections, per wave
( echo "# stem_rs_kernel at 256 pairs: cycles per tile (s_memtime) of its three sections per wave, option stem_dbg=9"
  echo "# -- stem_form=fast (the default: 209 MFMAs per wave and tile)"
  FORM=fast bash tools/stem_rs_prof.sh
  echo "# -- stem_form=resident (the tile kernel's summation order: 240 / 240 / 240 / 260 MFMAs)"
  FORM=resident bash tools/stem_rs_prof.sh ) > $O/${tag}_stem_rs_phases.txt 2>&1
# the row-streaming first-stage convs: role cycles per half-step + timing-only ablations (rebuilds conv_rows.o with the profiling code, restores)
( echo "# conv_rows32_kernel at 256 pairs: cycles per half-step of the two roles (s_memtime; profiling build), then ablations (WRONG results, timing only):"
  echo "# PNVO_ROWS_DBG bits: 1 no loads, 2 no stores, 4 no MFMAs, 8 no conversion; values = us per launch of the two GroupNorm-input convs"
  bash tools/rows_prof.sh ) > $O/${tag}_rows_phases.txt 2>&1
python tools/ab_option.py x3_rows off on > $O/${tag}_ab_x3_rows.txt 2>/dev/null
python tools/ab_option.py stem_form tiles fast > $O/${tag}_ab_stem_form.txt 2>/dev/null
python tools/ab_option.py stem_form tiles resident >> $O/${tag}_ab_stem_form.txt 2>/dev/null
python tools/ab_option.py stem_form tiles persistent >> $O/${tag}_ab_stem_form.txt 2>/dev/null
python tools/ab_option.py x3_persist off on > $O/${tag}_ab_x3_persist.txt 2>/dev/null
python tools/ab_dual.py x3_persist off on > $O/${tag}_ab_dual_persist.txt 2>/dev/null
python tools/bench_boundary.py > $O/${tag}_bench_boundary.json 2>/dev/null
python tools/bench_boundary_phases.py 64 > $O/${tag}_boundary_phases.json 2>/dev/null
python tools/bench_batch_sweep.py > $O/${tag}_batch_sweep.txt 2>/dev/null
python bench.py --config train --steps 10 --warmup 3 --no-overlap > $O/${tag}_bench_train_no_overlap.json 2>/dev/null
# one pair per call: the persistent small-batch kernel, its per-phase times, and the per-layer launches it replaces
# (traced with the plain launch: rocprofv3 7.2 crashes in its exit handler after a process used hipLaunchCooperativeKernel; the trace
#  itself is complete either way and the kernel time is the same)
PNVO_SMALL_COOP=0 rocprofv3 --kernel-trace --stats -d $O/${tag}_trace_b1 -o p -- python bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/${tag}_trace_b1.log 2>&1
python tools/rocprof_summary.py $O/${tag}_trace_b1/p_results.db > $O/${tag}_kernel_trace_b1.md 2>&1
PNVO_SMALL_NET=off rocprofv3 --kernel-trace --stats -d $O/${tag}_trace_b1l -o p -- python bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/${tag}_trace_b1l.log 2>&1
python tools/rocprof_summary.py $O/${tag}_trace_b1l/p_results.db > $O/${tag}_kernel_trace_b1_layers.md 2>&1
rm -rf $O/${tag}_trace_b1 $O/${tag}_trace_b1l
PROF=1 COOPS=0 python tools/check_small.py 2>&1 | grep -v amdgpu.ids > $O/${tag}_smallnet_phases.txt
# counters of the one-pair forward (the persistent kernel and the stem in front of it)
: > $O/${tag}_pmc_b1.md
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  PNVO_SMALL_COOP=0 rocprofv3 --kernel-trace --pmc $set -d $O/${tag}_pmc_b1_$n -o p -- python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-preheat --no-secondary > $O/${tag}_pmc_b1_$n.log 2>&1
  python tools/rocprof_summary.py $O/${tag}_pmc_b1_$n/p_results.db --pmc 2>&1 | awk '/## counters/{f=1} f' >> $O/${tag}_pmc_b1.md
  rm -rf $O/${tag}_pmc_b1_$n
done
head -14 $O/${tag}_kernel_trace_fwd_fp32.md
# navigation-loop proxy, its phases, the policy step alone, and the launches of small batches
python tools/bench_navloop.py --envs 8 16 32 --steps 100 > $O/${tag}_navloop.json 2>/dev/null
PNVO_NAVLOOP_PHASES=1 python tools/bench_navloop.py --envs 8 16 32 --steps 60 2>&1 >/dev/null | grep "\[navloop\]" > $O/${tag}_navloop_phases.txt
python tools/bench_policy.py --envs 1 4 8 16 32 > $O/${tag}_bench_policy.json 2>/dev/null
python tools/layer_breakdown.py 8 16 32 2>/dev/null > $O/${tag}_small_batch_breakdown.txt
python tools/grouped_breakdown.py 2>/dev/null > $O/${tag}_grouped_breakdown.txt
