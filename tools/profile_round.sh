cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r01l && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01l/trace -- python3 bench.py --serial --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01l/trace.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01l/pmc_fetch -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01l/pmc_write -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/pmc_write.log 2>&1 && \
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r01l/pmc_tcc -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/pmc_tcc.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r01l/pmc_sq -- python3 bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01l/pmc_sq.log 2>&1 && \
python tools/pmc_summary.py gpurun_out/r01l/summary.json gpurun_out/r01l/trace gpurun_out/r01l/pmc_fetch gpurun_out/r01l/pmc_write gpurun_out/r01l/pmc_tcc gpurun_out/r01l/pmc_sq > gpurun_out/r01l/summary.txt 2>&1 && \
python bench.py --steps 10 --warmup 3 > gpurun_out/r01l/line.json 2> gpurun_out/r01l/line.err; tail -3 gpurun_out/r01l/summary.txt; ls gpurun_out/r01l/trace/*/ | head; find gpurun_out/r01l/trace -name "*stats*" | head
