mkdir -p gpurun_out/r3d
for kv in 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv python profiles/timeline/timeline.py --envs 1 --steps 3 > gpurun_out/r3d/one_1e6_kernarg$kv.md 2>&1
  HIP_FORCE_DEV_KERNARG=$kv python profiles/timeline/timeline.py --envs 1 --particles 5000 --mesh 250 --steps 1 --calls 6 > gpurun_out/r3d/res_single_kernarg$kv.md 2>&1
  HIP_FORCE_DEV_KERNARG=$kv python profiles/pyloop_rl.py > gpurun_out/r3d/pyloop_kernarg$kv.log 2>&1
done
out=gpurun_out/r3d/ab; mkdir -p $out
cases=("one1e6|--steps 600 --warmup 60 --envs 1" "four1e6|--steps 400 --warmup 40 --envs 4" "env12|--steps 200 --warmup 20 --envs 12")
for r in 1 2; do for c in "${cases[@]}"; do tag=${c%%|*}; args=${c#*|}; for t in s8 noflush nostore sc1; do
  (cd profiles/ab/$t && python bench.py --no-cpu-baseline $args) > $out/${tag}_${t}_$r.json 2>/dev/null || echo "FAILED $tag $t"
done; done; done
for kv in 0 1; do (cd profiles/ab/s8 && HIP_FORCE_DEV_KERNARG=$kv python bench.py --no-cpu-baseline --steps 600 --warmup 60 --envs 1) > $out/one1e6_s8kernarg${kv}_1.json 2>/dev/null; (cd profiles/ab/s8 && HIP_FORCE_DEV_KERNARG=$kv python bench.py --no-cpu-baseline --steps 2000 --warmup 200 --envs 1 --particles 10000 --mesh 128) > $out/cfg1_s8kernarg${kv}_1.json 2>/dev/null; done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f"{os.path.basename(f):34s} {d['ms_per_step']*1e3:9.1f} us/step  " + str({n: round(v['avg_ms'] * 1e3, 1) for n, v in d['kernels'].items()}))
    except Exception as e:
        print(os.path.basename(f), "unreadable")
PY
grep -h "update_state(None), nothing\|raw handle" gpurun_out/r3d/pyloop_kernarg*.log
