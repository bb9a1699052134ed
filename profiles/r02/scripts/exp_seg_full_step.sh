for k in 64 4096; do for f in 0x180; do
  python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k $k --flags $f --iters 20 2>&1 | grep gathers | awk -v k=$k -v f=$f '{us=$0; sub(/.*: /,"",us); sub(/ us.*/,"",us); n=$0; sub(/.*\| /,"",n); sub(/ gathers.*/,"",n); printf "k=%d flags=%s: %s us -> %.1f TB/s\n", k, f, us, n*512/us/1e6}'
done; done
python profiles/r02/scripts/exp_seg_plan_widths.py 2>&1 | grep -v amdgpu | grep -E "products-sbm|com-amazon-sbm   N=(128|512)|com-amazon-like  N=(128|512)"
bash profiles/r02/scripts/exp_low_degree_floor.sh 2>&1 | grep -E "flags (0|0x80) " | cut -c1-60
python -m pytest tests/test_gpu_plan.py tests/test_gpu_spmm.py -x -q -m gpu 2>&1 | tail -2
