for d in 3 4 5 6 8; do for f in 0x20 0x80; do
  python profiles/r02/scripts/fetch_calibration.py --deg $d --flags $f --iters 200 2>&1 | grep gathers | cut -c1-50
done; done
for dd in "1,2,3,4,5,6,7" "2,4,6" "1,1,2,3,5,8,13"; do for f in 0x20 0x80; do
  python profiles/r02/scripts/fetch_calibration.py --degs $dd --flags $f --iters 200 2>&1 | grep gathers | cut -c1-60
done; done
