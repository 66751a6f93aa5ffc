cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s8; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/decode_variants.py --variants auto,vocab_generic,auto,vocab_generic > $O/variants.jsonl 2> $O/variants.err; cat $O/variants.jsonl; tail -3 $O/variants.err
python - <<'PY'
import sys; sys.path.insert(0,'.')
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
import numpy as np
d=PRESETS["large-v3-turbo"]
for B in (32, 48, 8):
    e=Engine(d, COMPUTE_BF16, B); e.load_weights(synth.iter_weights(d))
    e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False); e.encode(B); e.decode_reset(B)
    st=e.special
    a=e.decode_step([st.sot]*B)
    for v in (1,0,1,0):
        e.set_option("vocab_persistent", v)
        k=e.bench_kernel("logits_gemm", B, iters=200)
        print(B, "persistent" if v else "generic", round(k["ms"]*1e3,2), "us", round(k["bytes"]/k["ms"]/1e9,2), "TB/s")
    e.set_option("vocab_persistent", 0); e.decode_reset(B); b=e.decode_step([st.sot]*B)
    print("bit-identical logits:", np.array_equal(a,b), float(np.abs(a-b).max()))
    e.close()
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_batch.py tests/test_gpu_beam.py tests/test_gpu_align.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
