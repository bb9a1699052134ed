#!/bin/bash
# Round 5, call a: state of HEAD on a fresh box (GPU tests, bench line), the hit-bound hold-out graphs at N = 128 through every
# plan kernel, block shapes of the staged-rows kernel, and its phase clocks (instrumented build of the same sources).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.log
cp profiles/bench_extra_last.json $O/ 2>/dev/null
timeout 900 python scripts/kernel_ab.py --graphs geometric nws-k10 lfr-mu0.1 lfr-mu0.3 com-amazon-sbm com-amazon-like --widths 128 --kernels stream seg-stream staged --auto > $O/kernel_ab_n128.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 --kernels seg-stream staged > $O/kernel_ab_products.log 2>&1
for w in 8 4; do
  GESPMM_STAGED_WAVES=$w timeout 600 python scripts/kernel_ab.py --graphs geometric nws-k10 lfr-mu0.1 com-amazon-sbm --widths 128 --kernels staged --tag "waves=$w " >> $O/staged_waves.log 2>&1
done
GESPMM_STAGED_WAVES=8 timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 --kernels staged --tag "waves=8 " >> $O/staged_waves.log 2>&1
# phase clocks: the instrumented build in place of the library (this copy of the tree is scratch)
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_release.so
cp gespmm_amd/lib_instr/libgespmm.so gespmm_amd/lib/libgespmm.so
for w in 16 8; do
  GESPMM_STAGED_WAVES=$w GESPMM_STAGED_DEBUG=4 timeout 600 python scripts/kernel_ab.py --graphs geometric com-amazon-sbm nws-k10 --widths 128 --kernels staged --tag "clk waves=$w " >> $O/staged_clocks.log 2>&1
done
cp /tmp/libgespmm_release.so gespmm_amd/lib/libgespmm.so
tail -n 3 $O/pytest_gpu.log | cut -c1-300; tail -c 600 $O/bench.log; cat $O/kernel_ab_n128.log $O/kernel_ab_products.log $O/staged_waves.log | cut -c1-400; grep -v "^+" $O/staged_clocks.log | cut -c1-300
