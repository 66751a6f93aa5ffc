cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 200 python tools/beam_step_bench.py --clips 6,8 --beam 5 --new-tokens 32 2>/dev/null < /dev/null
timeout 200 python - <<'PY' < /dev/null
import sys; sys.path.insert(0,'.')
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
d=PRESETS["large-v3-turbo"]
for B in (40, 64, 96):
    e=Engine(d, COMPUTE_BF16, B); e.load_weights(synth.iter_weights(d))
    e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False); e.encode(B); e.decode_reset(B)
    e.decode_step([e.special.sot]*B)
    for name in ("dec_gemm_fc1",):
        k=e.bench_kernel(name, B, iters=200)
        print(B, name, round(k["ms"]*1e3,2), "us", flush=True)
    e.close()
PY
timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_wide_batch.py tests/test_gpu_beam.py -m gpu -q -x < /dev/null 2>&1 | tail -3
timeout 300 python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --max-clips 8 2>/dev/null < /dev/null
timeout 300 python tools/decode_variants.py --variants auto,auto 2>/dev/null < /dev/null
