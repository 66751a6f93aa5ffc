cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/beam_step_bench.py --clips 6,8 --beam 5 --new-tokens 32 2>/dev/null
python - <<'PY'
import sys; sys.path.insert(0,'.')
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
d=PRESETS["large-v3-turbo"]
for B in (32, 40, 64):
    e=Engine(d, COMPUTE_BF16, B); e.load_weights(synth.iter_weights(d))
    e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False); e.encode(B); e.decode_reset(B)
    e.decode_step([e.special.sot]*B)
    for name in ("dec_gemm_fc1","logits_gemm","xattn"):
        k=e.bench_kernel(name, B, iters=200)
        print(B, name, round(k["ms"]*1e3,2), "us", round(k["bytes"]/k["ms"]/1e9,2), "TB/s")
    e.close()
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -- python3 $GRAFT_REPO_ROOT/tools/beam_step_bench.py --clips 8 --beam 5 --new-tokens 32 > /dev/null 2>&1
f=$(find /tmp/bp -name '*kernel_stats.csv' | head -1); head -16 $f | cut -c1-200
