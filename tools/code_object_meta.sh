# Register metadata of the resident-weight kernels straight from the code objects (llvm-readelf --notes of the gfx950 bundle inside
# the .o files): what the rocprof summary's VGPR / AGPR columns cannot show for a unified register file.  No GPU needed.
#   bash tools/code_object_meta.sh > profiles/<tag>_code_object_registers.md
set -e
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
echo "| kernel | vgpr_count (unified, incl. AGPRs) | agpr_count | sgpr_count | sgpr spills | vgpr spills | scratch B/lane |"
echo "|---|---|---|---|---|---|---|"
for o in stem_rs conv_rows conv_x3; do
  cp pointnav-vo_amd/csrc/$o.o $T/ && (cd $T && $L/llvm-objdump --offloading $o.o > /dev/null 2>&1)
  $L/llvm-readelf --notes $T/$o.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | awk '
    /\.agpr_count:/ {a=$NF} /\.name:/ {n=$NF} /\.private_segment_fixed_size:/ {p=$NF} /\.sgpr_count:/ {s=$NF} /\.sgpr_spill_count:/ {ss=$NF}
    /\.vgpr_count:/ {v=$NF} /\.vgpr_spill_count:/ {print n, v, a, s, ss, $NF, p}' | grep -E "stem_rs_kernel|conv_rows32|conv_x3p_kernel" | while read n v a s ss vs p; do
      echo "| \`$(echo $n | c++filt | sed 's/void pnvo:://; s/(pnvo::.*//')\` | $v | $a | $s | $ss | $vs | $p |"; done
done
rm -rf $T
