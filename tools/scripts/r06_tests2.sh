R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
for i in 1 2 3; do timeout 600 python3 -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "interleaved" 2>&1 | tail -2; done > $O/r06_tests2a.log 2>&1
timeout 2400 python3 -m pytest tests -q -m gpu --durations=8 --deselect tests/test_lstm_gpu.py::test_lstm_graph_capture_after_an_interleaved_predict_repacks_weights > $O/r06_tests2.log 2>&1
cat $O/r06_tests2a.log; tail -15 $O/r06_tests2.log
