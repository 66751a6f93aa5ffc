cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/x.py <<'PY'
import sys; sys.path.insert(0,'.')
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
d=PRESETS["large-v3-turbo"]
for B in (40, 30, 10, 5):
    e=Engine(d, COMPUTE_BF16, B); e.load_weights(synth.iter_weights(d))
    A=B//5
    e.log_mel([synth.noise_clip(i) for i in range(A)], want_output=False); e.encode(A); e.decode_reset(B)
    for tgt in (480, 768, 1024, 1536, 2048, 480):
        e.set_option("xattn_mq_split_target", tgt)
        k=e.bench_kernel("xattn_beam5", B, iters=200)
        print(B, "rows target", tgt, round(k["ms"]*1e3,2), "us", flush=True)
    e.close()
PY
timeout 300 python /tmp/x.py < /dev/null
