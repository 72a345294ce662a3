cd $GRAFT_REPO_ROOT
for r in 1 2; do for gm in 6 5 7 12; do
  for w in flux qwen; do
    timeout 400 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-clip --no-wan --tune gemm.group_m=$gm 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'workload': '$w', 'gemm.group_m': $gm, 'round': $r, 'ms_per_step': round(d['ms_per_step'], 3)}))"
  done
done; done
