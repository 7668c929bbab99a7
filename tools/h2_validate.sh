#!/bin/bash
# round 4: the fp16x2 ("h2") default under the whole GPU suite + smoke + lane probes (one gpurun call)
mkdir -p gpurun_out/h2
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/h2/pytest_default.txt
tail -8 gpurun_out/h2/pytest_default.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/lane_race_probe.py 12 > gpurun_out/h2/lane_probe_bounded.txt 2>&1
IRR_LANE_MAX_LEAD=0 python tools/lane_race_probe.py 12 > gpurun_out/h2/lane_probe_unbounded.txt 2>&1
IRR_CONV_MATH=x3 IRR_LANE_MAX_LEAD=0 python tools/lane_race_probe.py 24 > gpurun_out/h2/lane_probe_unbounded_x3.txt 2>&1
python tools/truth_probe.py > gpurun_out/h2/truth_probe.txt 2>&1
grep -c "e-06" gpurun_out/h2/lane_probe_*.txt
