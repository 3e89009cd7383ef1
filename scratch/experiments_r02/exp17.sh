#!/bin/bash
# sample clocks / power while the benchmark runs (is the chip power- or clock-limited at full load?)
(python bench.py --no-cpu --no-single --no-verify --steps 40 > gpurun_out/exp17_bench.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|Power|GPU use|fclk" | head -8; echo --; sleep 0.5; done
wait $BP
tail -1 gpurun_out/exp17_bench.json | cut -c1-200
echo "== idle"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4
(TTN_WG512=0 python tools/diag_batch.py 1 > /dev/null 2>&1; for k in 1 2 3 4 5 6 7 8; do python - <<'PY'
PY
done) 
