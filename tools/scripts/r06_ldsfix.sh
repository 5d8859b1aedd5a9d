R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "bn_ or pool or stats or tiny_unet or full_unet_training_step or timed_configuration or fused_backward" > $O/r06_ldsfix_tests.log 2>&1
tail -3 $O/r06_ldsfix_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ldsfix_m
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/ldsfix_m -o m -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/ldsfix_m.log 2>&1
cd $R
python3 tools/pmc_mfma_summary.py $O/ldsfix_m $O/ldsfix_pmc_mfma.json 2>&1 | head -12
rm -rf $O/ldsfix_m
python3 tools/step_probe.py --only bn_bwd 2>&1 | tail -3
