for fl in "" "--no-graph"; do
 for rep in 1 2 3; do
  python bench.py --no-extras --no-cpu-baseline --no-converged --steps 20 --warmup 5 $fl 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1] or "graph", round(d["ms_per_step"]*1e3,3), "us/step; kernel_ms_per_launch", round(d["roofline"]["kernel_ms_per_launch"]*1e3,1))' "$fl"
 done
done
