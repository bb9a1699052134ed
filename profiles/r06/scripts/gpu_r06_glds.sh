#!/bin/bash
# Round 6: staged-rows kernel with direct-to-LDS staging (profiles/r06/experiments/staged_glds_build.py) against the product, interleaved.
export TMPDIR=/tmp
O=gpurun_out/r06glds; mkdir -p $O
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_product.so
for rep in 1 2 3; do
  for v in product glds; do
    if [ $v = product ]; then cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so; else cp profiles/r06/experiments/_build/glds/libgespmm.so gespmm_amd/lib/libgespmm.so; fi
    timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 512 --kernels staged --tag "$v " 2>&1 | grep -v amdgpu >> $O/staged_glds_ab.log
  done
done
cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so
cat $O/staged_glds_ab.log
