# exact HBM read bytes from the L2's request-size counters (cross-check of the FETCH_SIZE x 2 rule): calibration on the GEMM lab's known
# shapes, then the bench command on one stream.   usage: TAG=r05m bash tests/tools/pmc_exact.sh
T=${TAG:-pmcx}; R=$PWD
mkdir -p gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
C="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum"
timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/$T/lab_rd -o run -- $R/tests/tools/gemm_lab 1 shipped > $R/gpurun_out/$T/lab_rd.log 2>&1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/lab_fetch -o run -- $R/tests/tools/gemm_lab 1 shipped > $R/gpurun_out/$T/lab_fetch.log 2>&1
export STIL_WGRAD_STREAM=0
timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/$T/bench_rd -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/bench_rd.json 2> $R/gpurun_out/$T/bench_rd.err
cd $R
ls gpurun_out/$T
