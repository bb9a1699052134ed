#!/bin/bash
# GCN training epochs (config 4 shape: 2 convs, hidden 128) with and without the analysis stage.
cd $GRAFT_REPO_ROOT
for ds in com-amazon-sbm com-amazon-like pubmed; do
  for extra in "" "--no-plans"; do
    echo "== $ds $extra"
    python examples/gcn_custom.py --dataset $ds --n-hidden 128 --epochs 100 $extra 2>&1 | grep -v amdgpu | tail -2
  done
done
