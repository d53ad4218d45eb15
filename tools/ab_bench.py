"""GPU experiment helper: interleaved A/B/... runs of bench.py configurations (env-var sets), medians over several rounds, so
that clock / box drift does not decide the comparison.   usage: python tools/ab_bench.py ROUNDS "NAME:VAR=V,VAR=V:--flag x" ..."""
import json, os, statistics, subprocess, sys
rounds = int(sys.argv[1])
configs = []
for spec in sys.argv[2:]:
    name, envs, *flags = (spec.split(":") + ["", ""])[:3]
    env = dict(kv.split("=") for kv in envs.split(",") if kv)
    configs.append((name, env, flags[0].split() if flags and flags[0] else []))
res = {c[0]: [] for c in configs}
for r in range(rounds):
    for name, env, flags in configs:
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "1500", "--warmup", "100", "--preroll-min", "2048"] + flags,
                             env=dict(os.environ, **env), capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[name].append(d["value"] / 1e6)
for name, v in res.items():
    print(f"{name:28s} median {statistics.median(v):6.2f} M env-steps/s   all {[round(x, 1) for x in v]}")
