# N > 1 loop on one GPU through one-rank RCCL under rocprofv3 --kernel-trace: one steady-state step with RCCL's own kernels in it
# (host-bound under the profiler: durations and overlaps are what to read, not the step time); + the pass-B probe
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/n1_trace
UPSP_FORCE_COLLECTIVES=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/n1_trace -- python3 bench.py --force-chunked --defer-exchange --steps 6 --warmup 3 --no-cpu-baseline --no-reraycast > gpurun_out/n1_trace.log 2>&1
python3 tools/trace_step.py gpurun_out/n1_trace gather_pixel_rows_kernel > gpurun_out/n1_trace_step.txt 2>&1
find gpurun_out/n1_trace -name "*kernel_trace.csv" -size +20M -delete
cat gpurun_out/passb_probe.txt; tail -60 gpurun_out/n1_trace_step.txt
rm -rf gpurun_out/plain_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/plain_trace -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-reraycast > gpurun_out/plain_trace.log 2>&1
python3 tools/trace_step.py gpurun_out/plain_trace node_rows_kernel > gpurun_out/plain_trace_step.txt 2>&1
find gpurun_out/plain_trace -name "*kernel_trace.csv" -size +20M -delete
cat gpurun_out/plain_trace_step.txt
