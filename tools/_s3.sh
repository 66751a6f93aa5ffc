cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f16.py tests/test_gpu_weights_and_launch.py tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_large_width.py -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/bench_bf16.json'));print(d['value'],d['config']['phase_ms'],d['roofline']['frac'],d['mfma']['frac'],d['output_check']['crc_match'])"
timeout 600 python bench.py --no-cpu-baseline --compute f16 > $O/bench_f16.json 2>> $O/bench.err; echo "bench f16 rc=$?"; python -c "
import json;d=json.load(open('$O/bench_f16.json'));print(d['value'],d['config']['phase_ms'],d['roofline']['frac'],d['mfma']['frac'],d['output_check'])"
tail -5 $O/bench.err
