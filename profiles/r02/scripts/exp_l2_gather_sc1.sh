for k in 64 4096 16384; do for f in 0x180 0x8180 0x120 0x8120; do
  python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k $k --flags $f --iters 20 2>&1 | grep gathers | awk -v k=$k -v f=$f '{us=$0; sub(/.*: /,"",us); sub(/ us.*/,"",us); n=$0; sub(/.*\| /,"",n); sub(/ gathers.*/,"",n); printf "k=%d flags=%s: %s us -> %.1f TB/s\n", k, f, us, n*512/us/1e6}'
done; done
