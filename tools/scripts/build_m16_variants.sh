# profiling variants of the 16x16x32 tile only: recompile conv_igemm_m16.hip with the given -D flags and link it with the objects of the last
# full build.  Usage: build_m16_variants.sh "-DSATCV_M16_ABL=2" "-DSATCV_M16_PRIO=0" ...  -> libsatcv_m16<flags>.so
R=$(git rev-parse --show-toplevel)
P=$R/satellite_computervision_amd
for v in "$@"; do
  tag=$(echo "$v" | sed 's/-DSATCV_M16_//g' | tr -cd '[:alnum:]_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $v -c $P/csrc/conv_igemm_m16.hip -o /tmp/m16_$tag.o || exit 1
  objs=$(ls $P/csrc/_obj/*.o | grep -v conv_igemm_m16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libsatcv_m16$tag.so $objs /tmp/m16_$tag.o -ldl || exit 1
  echo built libsatcv_m16$tag.so
done
