#!/bin/bash
# (round 6; SUF and extra environment from the caller, e.g. SUF=_x SOME_KNOB=1)
# Calibrated PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately, MI355X_MICROARCH.md HBM section) on a 2-Gbase prefix of the
# bench recipe at the run's k / a -> gpurun_out/r06t$SUF/r06_traffic.json (copied to profiles/ by hand).  Run from the repo root on the GPU box.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06t$SUF
B="python bench.py --bases 2e9 --k 25 --a 22 --steps 1 --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0"
rocprofv3 --pmc FETCH_SIZE -d /tmp/cal_f -o cal -- tools/pmc_calib/pmc_calib > gpurun_out/r06t$SUF/calib_stdout.txt 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/cal_w -o cal -- tools/pmc_calib/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/run_f -o run -- $B > gpurun_out/r06t$SUF/pmc_fetch_bench.json 2> gpurun_out/r06t$SUF/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/run_w -o run -- $B > gpurun_out/r06t$SUF/pmc_write_bench.json 2> gpurun_out/r06t$SUF/pmc_write.err
CB=$(grep -o "bytes_per_kernel[ =:]*[0-9]*" gpurun_out/r06t$SUF/calib_stdout.txt | grep -o "[0-9]*$" | head -1)
python tools/pmc_traffic.py --calib-fetch $(find /tmp/cal_f -name "*.db" | head -1) --calib-write $(find /tmp/cal_w -name "*.db" | head -1) \
  --fetch $(find /tmp/run_f -name "*.db" | head -1) --write $(find /tmp/run_w -name "*.db" | head -1) --calib-bytes ${CB:-8589934592} \
  --command "$B" -o gpurun_out/r06t$SUF/r06_traffic.json > gpurun_out/r06t$SUF/pmc_traffic_summary.txt 2>&1
python tools/rocpd_pmc.py $(find /tmp/run_f -name "*.db" | head -1) > gpurun_out/r06t$SUF/pmc_fetch_2Gbases.txt 2>&1
python tools/rocpd_pmc.py $(find /tmp/run_w -name "*.db" | head -1) > gpurun_out/r06t$SUF/pmc_write_2Gbases.txt 2>&1
cat gpurun_out/r06t$SUF/pmc_traffic_summary.txt | head -20
