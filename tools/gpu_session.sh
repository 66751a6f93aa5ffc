set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_width.py tests/test_gpu_full_size.py tests/test_gpu_c5_c2.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
timeout 300 python - > $O/xkv.txt 2>&1 <<'PY'
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS["large-v3"]
e = Engine(dims, COMPUTE_BF16, 32); e.load_weights(synth.iter_weights(dims))
e.log_mel([synth.noise_clip(b) for b in range(32)], want_output=False)
for _ in range(4):
    e.encode(32); print(e.phase_ms(), flush=True)
e.close()
PY
cat $O/xkv.txt
