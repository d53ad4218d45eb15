cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-measure-traffic "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value']/1e6,2), d['roofline']['kernel_ms'], d['roofline'].get('frac_sustained'), d['config'].get('placement_probe_us'))"; }
run --device-warm-seconds $1
run --device-warm-seconds 2
run --device-warm-seconds 2
python bench.py --no-cpu-baseline --no-measure-traffic 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['value']/1e6,2), d['roofline']['kernel_ms'])"
