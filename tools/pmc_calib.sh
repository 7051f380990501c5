#!/bin/bash
# usage (on the GPU box): bash tools/pmc_calib.sh -- the two counter passes of tools/pmc_calib.hip -> gpurun_out/pmc_calibration.json
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
rm -rf gpurun_out/cal_f gpurun_out/cal_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/cal_f -o run --output-format csv -- ./halo2_prover_amd/build/pmc_calib > gpurun_out/cal_req.json 2> gpurun_out/cal_f_err.txt || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/cal_w -o run --output-format csv -- ./halo2_prover_amd/build/pmc_calib > /dev/null 2> gpurun_out/cal_w_err.txt || exit 1
python3 tools/pmc_calib.py gpurun_out/cal_f gpurun_out/cal_w gpurun_out/cal_req.json gpurun_out/pmc_calibration.json
rm -rf gpurun_out/cal_f gpurun_out/cal_w
