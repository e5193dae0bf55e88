"""Shared fixtures.  `-m "not gpu"` covers the oracle, the synthetic transmitter, the
host-side logic and that libdabgpu loads and exports the C ABI; `-m gpu` is parity proper
(HIP kernels through the C ABI vs the oracle)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def built():
    """Make sure libdabgpu.so and the oracle exist (hipcc cross-compiles on CPU-only hosts)."""
    import dabgpu
    if not os.path.exists(dabgpu.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    from oracle import oracle as O
    O.lib()
    return True


@pytest.fixture(scope="session")
def ensemble():
    from dabgpu import synth
    return synth.Ensemble(seed=0xDAB0, n_frames=5)


@pytest.fixture(scope="session")
def ensemble_iq(ensemble):
    return ensemble.iq()


def make_ctx(lane_mode=None, max_frames=64, ofdm_symbol_runs=0, unfused=False):
    """Context with the Viterbi variant pinned through dabgpu_cfg.flags: None = chosen by batch size (the default),
    1 = the codeword-per-lane kernels even for small batches, 0 = the wave-per-codeword kernels only."""
    import dabgpu
    flags = {None: 0, 0: dabgpu.FLAG_VITERBI_WAVE, 1: dabgpu.FLAG_VITERBI_LANE}[lane_mode]
    if unfused:
        flags |= dabgpu.FLAG_LANE_UNFUSED
    return _ordered_context()(device=0, max_frames=max_frames, flags=flags, ofdm_symbol_runs=ofdm_symbol_runs)


_ORDERED = None


def _ordered_context():
    """dabgpu.Context whose device-pointer calls are ordered behind the HARNESS's producer, and nothing else.  The library's
    own stream is non-blocking: a tensor a test has just created (its zero-fill, its random numbers) is written on torch's
    current stream and nothing orders a `_dev` call behind it (small tensors are done long before Python gets to the call --
    but that is luck; tools/decoder_fuzz.py lost that race with a large one).  So before each `_dev` call an event is
    recorded on torch's current stream and the stream the call will run on -- the context's own, or the one the test
    passes -- waits for it ON THE DEVICE.  No host synchronisation: two library calls a test issues in a row reach the
    device back to back, so the ordering the library itself must provide between its calls (state events across caller
    streams, shared scratch, ring hand-over) is still what the tests exercise (the round-4 wrapper called
    torch.cuda.synchronize() here and hid all of that)."""
    global _ORDERED
    if _ORDERED is None:
        import inspect
        import sys
        import dabgpu

        class OrderedContext(dabgpu.Context):
            def __getattribute__(self, name):
                attr = super().__getattribute__(name)
                if name.endswith("_dev") and callable(attr):
                    torch = sys.modules.get("torch")
                    if torch is not None and torch.cuda.is_available():
                        def ordered(*a, **kw):
                            ev = torch.cuda.current_stream().record_event()
                            target = None
                            try:
                                target = inspect.signature(attr).bind(*a, **kw).arguments.get("stream")
                            except TypeError:
                                pass
                            torch.cuda.ExternalStream(target if target else super(OrderedContext, self).__getattribute__("stream")).wait_event(ev)
                            return attr(*a, **kw)
                        return ordered
                return attr

        _ORDERED = OrderedContext
    return _ORDERED


@pytest.fixture(scope="session", params=["auto", "lane"])
def ctx(built, request):
    """Every parity test runs twice: with the default kernel selection (small test batches -> one wavefront per
    codeword) and with the large-batch codeword-per-lane Viterbi forced, so both are held to the oracle."""
    c = make_ctx(1 if request.param == "lane" else None)
    yield c
    c.close()


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
