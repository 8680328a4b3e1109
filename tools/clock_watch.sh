# engine clock and package power while the two hot kernels run (sustained cfg3 steps): is the part power-limited?
# usage (GPU box): bash tools/clock_watch.sh  -> gpurun_out/clock_watch.txt
mkdir -p gpurun_out
OUT=gpurun_out/clock_watch.txt
: > $OUT
rocm-smi --showclocks --showpower --showtemp -d 0 >> $OUT 2>&1
python3 bench.py --config cfg3 --steps 600 --warmup 5 --no-cpu-baseline > gpurun_out/clock_watch_bench.json 2>/dev/null &
BP=$!
sleep 4
for i in 1 2 3 4 5 6; do
  echo "--- sample $i (bench running)" >> $OUT
  rocm-smi --showclocks --showpower -d 0 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|power" >> $OUT
  sleep 0.7
done
wait $BP
tail -1 gpurun_out/clock_watch_bench.json | cut -c1-160 >> $OUT
echo "--- idle again" >> $OUT
rocm-smi --showclocks --showpower -d 0 2>&1 | grep -E "sclk|mclk|Power|power" >> $OUT
cat $OUT
