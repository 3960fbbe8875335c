import sys, json, subprocess
for i in range(3):
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "100", "--warmup", "20", "--steady-steps", "0"], capture_output=True, text=True)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(round(d["ms_per_step"] * 1e3, 1), d["placement"], flush=True)
