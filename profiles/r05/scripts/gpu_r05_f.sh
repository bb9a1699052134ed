#!/bin/bash
# Round 5, call f: full GPU suite and the bench line with the record-stream kernel and the cost-aware plans; GCN epochs with the
# reference's cached=True (pubmed: the plan must not cost more than it returns); the analysis stage phase by phase.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.log
cp profiles/bench_extra_last.json $O/ 2>/dev/null
for ds in pubmed com-amazon-sbm com-amazon-like; do
  for extra in "" "--no-plans"; do
    echo "== $ds hidden 128 epochs 100 $extra" >> $O/gcn_epochs.log
    timeout 600 python examples/gcn_custom.py --dataset $ds --n-hidden 128 --epochs 100 $extra 2>&1 | grep -v "amdgpu\|^W2026" | tail -2 >> $O/gcn_epochs.log
  done
done
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/kernel_pmc_case.py com-amazon-sbm 128 auto 2 > $O/plan_phases.log 2>&1
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/kernel_pmc_case.py pubmed-like 128 stream 2 >> $O/plan_phases.log 2>&1
cat $O/gcn_epochs.log; grep "^{" $O/bench.log | cut -c1-1500; grep "\[plan\]" $O/plan_phases.log | head -80
