set -x
O=gpurun_out/r05g; mkdir -p $O
bash profiles/solo_trace.sh r05g > /dev/null 2>&1; cp gpurun_out/r4/solo_kernel_stats_r05g.csv $O/kernel_stats_single_stream.csv
CLMGS_LIB_PATH=$PWD/clm_gs_amd/libclmgs_hip_prof.so CLMGS_BINNING=r4 bash profiles/solo_trace.sh r05g_r4 > /dev/null 2>&1; cp gpurun_out/r4/solo_kernel_stats_r05g_r4.csv $O/kernel_stats_single_stream_r4route.csv
grep -E "emit|hist|scatter|scan|count_lb|keys_lb|offsets" $O/kernel_stats_single_stream.csv
echo ----
grep -E "emit|hist|scatter|scan|count_lb|keys_lb|offsets" $O/kernel_stats_single_stream_r4route.csv
