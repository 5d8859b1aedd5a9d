# rocprofv3 kernel statistics of the ConvLSTM family (tools/lstm_time.py: training step + inference pass of get_lstm_model, get_lstm_autoencoder,
# get_hybrid_model, get_hierarchical_model at the generators' default shapes) -> gpurun_out/lstm_kernel_stats.csv, lstm_times.txt
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/lstm_raw
timeout 600 python3 $R/tools/lstm_time.py > $O/lstm_times.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lstm_raw -o l -- python3 $R/tools/lstm_time.py > $O/lstm_trace.log 2>&1
cp $(find $O/lstm_raw -name "*kernel_stats.csv" | head -1) $O/lstm_kernel_stats.csv 2>/dev/null
rm -rf $O/lstm_raw
cat $O/lstm_times.txt | grep -v amdgpu.ids; head -25 $O/lstm_kernel_stats.csv | cut -c1-160
