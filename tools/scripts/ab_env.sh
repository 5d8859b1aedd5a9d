# A/B of environment switches on one box, one after the other, twice (A B C A B C): usage ab_env.sh "VAR=1" "VAR=2 OTHER=3" ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
for rep in 1 2; do
  for cfg in "$@"; do
    out=$(env $cfg python3 bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$cfg :: $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'tiles/s frac', d['roofline']['frac'], d['extra']['kernel_ms_per_step'])")"
  done
done
