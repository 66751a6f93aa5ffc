import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the session's weight cache hands out read-only arrays; torch.from_numpy mentions it once per call site
    config.addinivalue_line("filterwarnings", "ignore:The given NumPy array is not writable")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _oracle_thread_count():
    """The CPU oracle (torch) is the host-time cost of the GPU suite.  On the GPU box (256 logical CPUs) torch defaults to 128
    threads, and its many small ops run ~3x SLOWER that way than with 32 (measured round 4: oracle encoder over 16 clips at
    large-v3 width 12.9 s -> 4.7 s, an 8-token greedy decode 4-6 s -> 1-1.7 s; bench.py's cpu_baseline caps at 32 for the same
    reason).  VERDICT round 3, next #6."""
    try:
        import torch
        n = os.cpu_count() or 1
        if n > 32:
            torch.set_num_threads(32)
    except Exception:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _synthetic_weight_cache():
    """The GPU suite builds the same seeded synthetic tensors many times (whisper-large-v3 alone - 1.5 B values, ~10 s of Philox
    per generation on the GPU box's host - for eight engines / oracle copies): memoise synth.make_tensor for the session.  Cached
    arrays are read-only, so a test that tried to edit one in place fails loudly instead of corrupting its neighbours
    (consumers copy: .astype / torch.from_numpy(...).to / the engine's staging upload).  VERDICT round 3, next #6."""
    from taiwan_tongues_asr_ce_amd import synth
    orig, cache = synth.make_tensor, {}

    def cached(name, shape, kind, seed=0, profile="gauss"):
        key = (name, tuple(shape), kind, seed, profile)
        a = cache.get(key)
        if a is None:
            a = orig(name, shape, kind, seed, profile)
            a.flags.writeable = False
            cache[key] = a
        return a
    synth.make_tensor = cached
    yield
    synth.make_tensor = orig
    cache.clear()
