# samples the shader clock while the dc_gan bench runs (is the fp32 MFMA peak of 157 TFLOP/s at 2.4 GHz reachable?)
cd $GRAFT_REPO_ROOT
python bench.py --steps 2500 --warmup 5 --reps 1 --no-cpu-baseline --no-bs128 --no-kernel-timer > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done > gpurun_out/clock_samples.txt
wait $BP
