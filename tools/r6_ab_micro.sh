#!/bin/bash
# round 6: per-kernel A/B of this tree against the end-of-round-5 tree (r5tree/, git archive of 82f2c7e with its own library; not tracked):
# the split-operand forward / weight-gradient shapes of tools/h2_check.py (speed part only), alternating.   bash tools/r6_ab_micro.sh [rounds]
R=${1:-2}
for i in $(seq 1 $R); do
  (cd r5tree && python tools/h2_check.py --noacc 2>/dev/null | grep -v "^==\|amdgpu" | sed 's/^/r5 /')
  python tools/h2_check.py --noacc 2>/dev/null | grep -v "^==\|amdgpu" | sed 's/^/r6 /'
done
