for c in "--deg 1 --identity" "--deg 1" "--deg 2" "--deg 3" "--degs 1,1,1,2,2,3,4,8" "--graph com-amazon-like" "--graph com-amazon-sbm"; do
 for f in 0 0x20 0x80 0x30 0x90; do
  python profiles/r02/scripts/fetch_calibration.py $c --flags $f --iters 200 2>&1 | grep gathers | cut -c1-60
 done
done
