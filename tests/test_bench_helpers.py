"""CPU: bench.py's token-agreement report (`output_check.vs_f32_parity_tokens`) on the committed f32 reference tokens and margins
(profiles/bench_tokens_f32.npy, _margins.npy, _runner_up.npy): identical tokens -> 32 / 32; a row that takes the f32 runner-up at a
position is reported with the f32 engine's own margin there and `took_ref_runner_up`; a row that takes some third token is not."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_token_agreement_reports_margins_at_divergences():
    b = _bench()
    ref = np.load(os.path.join(ROOT, "profiles", "bench_tokens_f32.npy"))
    mg = np.load(os.path.join(ROOT, "profiles", "bench_tokens_f32_margins.npy"))
    ru = np.load(os.path.join(ROOT, "profiles", "bench_tokens_f32_runner_up.npy"))
    assert ref.shape == mg.shape == ru.shape == (32, 128) and (mg > 0).all()        # the f32 tokens ARE the f32 argmax everywhere
    same = b.token_agreement(ref.copy(), "bench_tokens_f32.npy", "large-v3", "noise")
    assert same["rows_identical"] == 32 and same["equal_prefix_fraction"] == 1.0 and same["divergences"] == []
    mine = ref.copy()
    mine[3, 2:] = 7                      # leaves at position 2 ...
    mine[3, 2] = ru[3, 2]                # ... by taking the f32 runner-up
    mine[9, 40:] = 11                    # leaves at position 40 with a third token
    assert ru[9, 40] != 11
    out = b.token_agreement(mine, "bench_tokens_f32.npy", "large-v3", "noise")
    assert out["rows_identical"] == 30 and out["first_divergence_per_row"][3] == 2 and out["first_divergence_per_row"][9] == 40
    d = {x["row"]: x for x in out["divergences"]}
    assert set(d) == {3, 9}
    assert d[3]["took_ref_runner_up"] and abs(d[3]["ref_engine_top2_margin"] - float(mg[3, 2])) < 1e-4
    assert not d[9]["took_ref_runner_up"] and abs(d[9]["ref_engine_top2_margin"] - float(mg[9, 40])) < 1e-4
    assert abs(out["largest_ref_margin_at_a_divergence"] - max(float(mg[3, 2]), float(mg[9, 40]))) < 1e-4
    # another workload (other model / clip set) has no committed reference: no claim is made
    assert b.token_agreement(mine, "bench_tokens_f32.npy", "small", "noise") is None
    assert b.token_agreement(mine, "bench_tokens_f32.npy", "large-v3", "tonal") is None
