# profiling variants of the persistent 16x16x32 kernel only: recompile conv_igemm_m16p.hip with the given -D flags and link it with the objects of
# the last full build.  Usage: build_m16p_variants.sh "-DSATCV_M16P_ABL=2" ...  -> libsatcv_m16p<flags>.so
R=$(git rev-parse --show-toplevel)
P=$R/satellite_computervision_amd
for v in "$@"; do
  tag=$(echo "$v" | sed 's/-DSATCV_M16P_//g' | tr -cd '[:alnum:]_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $v -c $P/csrc/conv_igemm_m16p.hip -o /tmp/m16p_$tag.o || exit 1
  objs=$(ls $P/csrc/_obj/*.o | grep -v conv_igemm_m16p.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libsatcv_m16p$tag.so $objs /tmp/m16p_$tag.o -ldl || exit 1
  echo built libsatcv_m16p$tag.so
done
