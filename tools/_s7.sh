cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_c5_c2.py tests/test_gpu_facade.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 300 python bench.py --model large-v3-turbo --batch 32 --no-cpu-baseline > $O/bench_turbo.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_turbo.json'));print('turbo',d['value'],d['config']['phase_ms'])"
timeout 300 python bench.py --model small --batch 8 --no-cpu-baseline > $O/bench_small.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_small.json'));print('small',d['value'],d['config']['phase_ms'])"
timeout 400 python bench.py --new-tokens 444 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_444.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_444.json'));print('444',d['value'],d['config']['phase_ms'])"
timeout 400 python bench.py --more-in-flight --no-cpu-baseline > $O/bench_mif.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_mif.json'));print('mif',d['value'],d['more_in_flight'])"
