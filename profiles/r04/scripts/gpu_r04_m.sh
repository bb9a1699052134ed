#!/bin/bash
# Round 4: config 4 (GCN through the op) on the final library, with and without plans, eager and HIP-graph replay.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{
for ds in pubmed com-amazon-sbm com-amazon-like reddit-like reddit-sbm; do
  ep=100; [ "$ds" = "reddit-like" -o "$ds" = "reddit-sbm" ] && ep=20
  for extra in "" "--no-plans" "--graph-capture"; do
    echo "== $ds hidden 128 epochs $ep $extra"
    timeout 600 python examples/gcn_custom.py --dataset $ds --n-hidden 128 --epochs $ep $extra 2>&1 | grep -v "amdgpu\|^W2026" | tail -2
  done
done
} > gpurun_out/r04/gcn_epochs.log 2>&1
cat gpurun_out/r04/gcn_epochs.log | cut -c1-250
