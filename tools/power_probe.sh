# Not a test: samples rocm-smi (shader clock, socket power) while bench.py's step loop runs for a few seconds (GPU box, repo root)
python bench.py --steps 600 --warmup 4 --no-cpu-baseline --no-alt-schedule --no-extras > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | grep "GPU\[0\]" | tr '\n' ' '; echo; sleep 0.1; kill -0 $BP 2>/dev/null || break; done > gpurun_out/power_probe.txt
wait $BP
python - <<PY
import json
d=json.loads(open("gpurun_out/power_bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["stages_ms"])
PY
grep -vE "\((1[0-9][0-9]|9[0-9])Mhz\)" gpurun_out/power_probe.txt | sed 's/GPU\[0\]\t\t: //g' | tail -40
