for m in 0 1 2 4 8 16 3 15; do
  echo -n "skip=$m: "; TTASR_SKIP=$m timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['phase_ms'])"
done
