#!/bin/bash
# Round 6: cache-policy bits of the staged-rows kernel's loads on top of `sc1 nt` stores, interleaved.
export TMPDIR=/tmp
O=gpurun_out/r06loadpol; mkdir -p $O
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_product.so
for rep in 1 2 3; do
  for v in product Gnt Gsc1 Snt GntSnt; do
    if [ $v = product ]; then cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so; else cp profiles/r06/experiments/_build/load_$v/libgespmm.so gespmm_amd/lib/libgespmm.so; fi
    timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm products-sbm --widths 128 512 --kernels staged --tag "$v " 2>&1 | grep -v amdgpu >> $O/staged_load_policy.log
  done
done
cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so
sort -s -k2,3 $O/staged_load_policy.log
