#!/bin/bash
# Round 4: full GPU suite + the default bench run on the current tree.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke_e.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r04/smoke_e.log
timeout 2400 python -m pytest tests -m gpu -q -rs > gpurun_out/r04/pytest_gpu_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_e.log
( time python bench.py > gpurun_out/r04/bench_e.log 2> gpurun_out/r04/bench_e.err ) 2> gpurun_out/r04/bench_e.time
tail -2 gpurun_out/r04/smoke_e.log | cut -c1-300; tail -6 gpurun_out/r04/pytest_gpu_e.log | cut -c1-300; tail -1 gpurun_out/r04/bench_e.log | cut -c1-4000; cat gpurun_out/r04/bench_e.time
