#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python scripts/microbench.py > gpurun_out/microbench.log 2>&1
timeout 900 python scripts/sweep.py --quick --graphs com-amazon-like,com-amazon-like@0.9 --ncols 128 --rounds 2 > gpurun_out/sweep3.log 2>&1
timeout 900 python scripts/ksweep.py > gpurun_out/ksweep2.log 2>&1
timeout 300 python examples/gcn_custom.py --n-hidden 128 --epochs 200 > gpurun_out/gcn_pubmed.log 2>&1
timeout 300 python examples/gcn_custom.py --n-hidden 128 --epochs 200 --graph-capture > gpurun_out/gcn_pubmed_graph.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
