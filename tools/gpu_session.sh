set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2r; mkdir -p $O
timeout 300 python - > $O/flash_dbg.txt 2>&1 <<'PY'
import ctypes, json
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS["large-v3"]
e = Engine(dims, COMPUTE_BF16, 32); e.load_weights(synth.iter_weights(dims))
e.log_mel([synth.noise_clip(b) for b in range(32)], want_output=False); e.encode(32)
var = ctypes.c_int.in_dll(e.lib, "g_flash_dbg")
names = {0: "real", 1: "no global loads / LDS stores in the loop", 2: "no exp (add only)", 3: "no PV MFMAs (sum MFMA only)", 4: "real, 2 WG/CU", 5: "real, 1 WG/CU"}
for rep in range(2):
    for v in range(6):
        var.value = v
        k = e.bench_kernel("enc_attn", 32, iters=20)
        print(json.dumps({"dbg": v, "what": names[v], "us": round(k["ms"]*1e3, 1)}), flush=True)
e.close()
PY
cat $O/flash_dbg.txt
