cd $GRAFT_REPO_ROOT
for r in 96 104 108 109 110 112 116 120 124 131; do
  echo -n "rows=$r: "; GESPMM_STAGED_ROWS=$r timeout 200 python scripts/kernel_ab.py --graphs com-amazon-sbm --widths 128 --kernels staged 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
done
for r in 96 104 108 109 110 112 116 120 124 131; do
  echo -n "rows=$r: "; GESPMM_STAGED_ROWS=$r timeout 200 python scripts/kernel_ab.py --graphs com-amazon-sbm --widths 128 --kernels staged 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
done
