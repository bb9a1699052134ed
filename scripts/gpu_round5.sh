#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_driver.py -m gpu -x -q > gpurun_out/pytest_gpu_driver.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_driver.log
timeout 600 python scripts/microbench.py > gpurun_out/microbench.log 2>&1
timeout 300 python examples/gcn_custom.py --n-hidden 128 --epochs 200 --graph-capture > gpurun_out/gcn_pubmed_graph.log 2>&1
timeout 300 python examples/gcn_custom.py --n-hidden 128 --epochs 200 --convs 3 > gpurun_out/gcn_pubmed_3conv.log 2>&1
tail -3 gpurun_out/pytest_gpu_driver.log
