#!/bin/bash
# After a change to the analysis kernels: device order == host order (tests + fuzz), plan products == plain call, then the timings.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/plan_check; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_plan_device.py tests/test_gpu_plan.py -x -q > $O/pytest_plan.log 2>&1; echo "pytest rc=$?" >> $O/pytest_plan.log
timeout 600 python scripts/plan_device_fuzz.py 150 3 > $O/fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz.log
bash scripts/plan_profile_r06.sh > $O/profile.log 2>&1
python scripts/plan_ms.py --reps 5 products-sbm reddit-sbm com-amazon-like > $O/plan_ms_others.log 2>&1
tail -3 $O/pytest_plan.log; tail -3 $O/fuzz.log; tail -2 $O/profile.log; tail -4 $O/plan_ms_others.log | cut -c1-200
