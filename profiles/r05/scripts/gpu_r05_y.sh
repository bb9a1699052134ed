#!/bin/bash
# Round 5, call y: GPU suite after the narrow-width keep rule moved to a share of 0.80; default-life against steady-state plans on every
# hold-out graph at N = 32 / 64 / 128 (reorder = AUTO here: what a caller with the defaults gets).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 2400 python scripts/plan_life_compare.py --graphs ba-m6 geometric holme-kim-m5 lfr-mu0.1 lfr-mu0.3 lfr-mu0.5 nws-k10 com-amazon-sbm products-sbm --widths 32 64 128 2>&1 | grep -v amdgpu > $O/plan_life_compare_all.log
cat $O/plan_life_compare_all.log
