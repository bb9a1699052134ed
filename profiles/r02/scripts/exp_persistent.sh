#!/bin/bash
cd $GRAFT_REPO_ROOT
python profiles/r02/scripts/exp_persistent.py 2>&1 | grep -v amdgpu
for w in 128 64 32; do GESPMM_PERSIST_WGS=$w python profiles/r02/scripts/exp_persistent.py com-amazon-sbm 2>&1 | grep -v amdgpu | grep persistent; done
