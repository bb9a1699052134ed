#!/bin/bash
# Round 5, call x: after the depth of the analysis became width-aware (launches x N >= 100 000) and the narrow-width tables are built
# from 0.65 modelled hits: GPU suite, the life comparison again, the driver protocol.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05x; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 2400 python scripts/plan_life_compare.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 --widths 32 64 128 512 2>&1 | grep -v amdgpu > $O/plan_life_compare.log
cat $O/plan_life_compare.log
timeout 1200 python scripts/driver_compare.py com-amazon-sbm 2>&1 | grep -v amdgpu > $O/driver_compare.log
cat $O/driver_compare.log
