#!/bin/bash
# round 6, GPU batch 1: new tests, PMC passes for the default-path kernels, the C.3 wait-state probe, one bench line
mkdir -p gpurun_out/b1
python -m pytest tests/test_ddp_gpu.py tests/test_ops_gpu.py tests/test_harness_cpu.py "tests/test_train_gpu.py::test_forward_inside_a_running_backward_keeps_the_live_pass" "tests/test_train_gpu.py::test_zero_grad_after_forward_and_failed_backward_under_the_auto_lane" tests/test_h2_gpu.py -x -q > gpurun_out/b1/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/b1/pytest.txt
tail -5 gpurun_out/b1/pytest.txt
bash tools/r6_pmc.sh r6 > gpurun_out/b1/pmc.log 2>&1
FORMS=0,2,15,16,17,18,19,20,21,22,23,1 KINDS="alone,synthetic MFMA kernel,d16 weight gradient" timeout 600 python tools/pkfma_swap.py > gpurun_out/b1/r6_pkfma_waitstates.txt 2>&1
tail -40 gpurun_out/b1/r6_pkfma_waitstates.txt
timeout 900 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/b1/bench.json 2> gpurun_out/b1/bench.err
tail -c 1500 gpurun_out/b1/bench.json
