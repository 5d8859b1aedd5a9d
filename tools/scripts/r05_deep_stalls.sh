# Isolated-launch counter attribution of the deep 3x3 tile (review item 1a): separate --pmc passes of tools/conv_probe.py, the
# interpreter itself after `--`.  Usage: bash tools/scripts/r05_deep_stalls.sh <tag> [extra conv_probe args]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
TAG=${1:-deep}; shift
cd /tmp && export TMPDIR=/tmp
SH="--shapes 64,16,16,1024,512 64,64,64,256,128 64,32,32,512,256 --affine --reps 10"
rm -rf $O/st_${TAG}_a $O/st_${TAG}_b $O/st_${TAG}_c
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/st_${TAG}_a -o a -- python3 $R/tools/conv_probe.py $SH "$@" > $O/st_${TAG}_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/st_${TAG}_b -o b -- python3 $R/tools/conv_probe.py $SH "$@" > $O/st_${TAG}_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/st_${TAG}_c -o c -- python3 $R/tools/conv_probe.py $SH "$@" > $O/st_${TAG}_c.log 2>&1
cd $R
python3 tools/pmc_kernel_summary.py $O/r05_${TAG}_tile_stalls.json $O/st_${TAG}_a $O/st_${TAG}_b $O/st_${TAG}_c --match igemm > $O/r05_${TAG}_tile_stalls.txt 2>&1
rm -f $O/st_${TAG}_*/*/*kernel_trace.csv
tail -40 $O/r05_${TAG}_tile_stalls.txt
