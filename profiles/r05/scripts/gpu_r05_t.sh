#!/bin/bash
# Round 5, call t: fuzz of the reworked analysis stage (device order == host order, task tables, SpMM bits: tiny / rectangular / hub /
# duplicate-entry matrices) and the general soak of every launch form, new seeds.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
timeout 1500 python scripts/plan_device_fuzz.py 400 501 2>&1 | grep -v amdgpu | tail -8 > $O/plan_device_fuzz.log
timeout 1500 python scripts/plan_device_fuzz.py 400 502 2>&1 | grep -v amdgpu | tail -8 >> $O/plan_device_fuzz.log
timeout 2400 python scripts/soak_fuzz.py --cases 1500 --seed 77 2>&1 | grep -v amdgpu | tail -8 > $O/soak_fuzz.log
cat $O/plan_device_fuzz.log $O/soak_fuzz.log
