mkdir -p gpurun_out/compact
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "brick or composited or trained_like or render or march or frames" > gpurun_out/compact/parity.log 2>&1; tail -3 gpurun_out/compact/parity.log
for cfg in "--config 1" "--config 2 --sample-res 64,64,1,1,1,1" "--config 4 --slice-of 64" "--scene shopping_big --sample-res 32,32,1,1,1,1"; do
  echo "== $cfg"
  BENCH_ARGS="--clip vit_tiny $cfg" STEPS=4 ROUNDS=2 bash tools/opt_ab.sh "march_compact=0" "march_compact=1" "march_compact=1 --opt refill_min=32" "march_compact=1 --opt refill_min=48"
done > gpurun_out/compact/ab.log 2>&1
cat gpurun_out/compact/ab.log
