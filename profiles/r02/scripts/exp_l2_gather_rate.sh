# gather rate when every B row is an L2 hit: 232965 rows x 64 non-zeros, columns drawn from k rows of B (k x 512 B)
for k in 1024 2048 4096 6144 8192 12288 16384 32768; do
  for f in 0x20 0x30 0x80; do
    python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k $k --flags $f --iters 20 2>&1 | grep gathers | awk -v k=$k '{us=$0; sub(/.*: /,"",us); sub(/ us.*/,"",us); n=$0; sub(/.*\| /,"",n); sub(/ gathers.*/,"",n); printf "k=%d (%.1f MB of B) flags=%s: %s us -> %.1f TB/s of gathers\n", k, k*512/1e6, "'$f'", us, n*512/us/1e6}'
  done
done
