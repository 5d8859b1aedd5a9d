R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $O/wgpmc$v
  SATCV_WGRAD_M16=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/wgpmc$v -o m -- python3 $R/tools/wgrad_probe.py > $O/wgpmc$v.log 2>&1
  python3 $R/tools/pmc_kernel_summary.py $O/wgpmc$v.json $O/wgpmc$v --match wgrad_dma 2>&1 | cut -c1-600 | head -6
  rm -rf $O/wgpmc$v
done
