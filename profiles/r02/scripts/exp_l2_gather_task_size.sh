for k in 64 4096; do for f in 0x180 0x120; do for rpw in 0 2 4 8 16 32; do
  python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k $k --flags $f --rpw $rpw --iters 20 2>&1 | grep gathers | awk -v k=$k -v f=$f -v r=$rpw '{us=$0; sub(/.*: /,"",us); sub(/ us.*/,"",us); n=$0; sub(/.*\| /,"",n); sub(/ gathers.*/,"",n); printf "k=%d flags=%s rows_per_wave=%d: %s us -> %.1f TB/s\n", k, f, r, us, n*512/us/1e6}'
done; done; done
