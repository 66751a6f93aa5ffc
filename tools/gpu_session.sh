set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2u; mkdir -p $O
tools/microbench/bin/red_test > $O/red_test.txt 2>&1; echo "red_test rc=$?"; cat $O/red_test.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_width.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc=$?"; tail -3 $O/pytest_a.log
for rep in 1 2; do
timeout 300 python tools/decode_variants.py --variants auto > $O/new_$rep.txt 2>&1; cat $O/new_$rep.txt
timeout 300 python scratch_ab/old/tools/decode_variants.py --variants auto > $O/old_$rep.txt 2>&1; cat $O/old_$rep.txt
done
timeout 300 python - > $O/kern.txt 2>&1 <<'PY'
import json
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS["large-v3"]
e = Engine(dims, COMPUTE_BF16, 32); e.load_weights(synth.iter_weights(dims))
e.log_mel([synth.noise_clip(b) for b in range(32)], want_output=False); e.encode(32)
for name in ("xattn", "enc_attn"):
    for rep in range(3):
        k = e.bench_kernel(name, 32, iters=48)
        print(json.dumps({"kernel": name, "us": round(k["ms"]*1e3, 2)}), flush=True)
for _ in range(2):
    e.encode(32); print(e.phase_ms(), flush=True)
e.close()
PY
cat $O/kern.txt
