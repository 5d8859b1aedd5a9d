R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
for v in "" PRIO3 ABL16 ABL8 ABL64; do
 for s in "64,64,64,128,256 --affine" "64,128,128,64,128"; do echo "== variant=${v:-stamp} $s"; SATCV_LIB=$R/satellite_computervision_amd/libsatcv_m16pDSATCV_STAMP_M16P$v.so python3 tools/m16p_stamp_probe.py $s 2>&1 | grep -A12 "workgroup 0" | grep "wave  [048]"; done
done > $O/r06_m16p_stamp_variants.txt 2>&1
cat $O/r06_m16p_stamp_variants.txt
