"""Randomised parity sweep and the running-inverse stress cases as -m gpu tests (they were builder-run tools in round 2):
a fixed-seed slice of tests/_fuzz.py's cases -- every kernel family, wave organisation and fusion form against the CPU
oracle at 1e-8 -- and the long-horizon lambda extremes of tools/ns_stress.py."""
import numpy as np
import pytest

from oracle import pipeline
from trajoptkp_amd import Engine, synth

import _fuzz

pytestmark = pytest.mark.gpu

FUZZ_SEED, FUZZ_CASES, FUZZ_SLICES = 20261004, 220, 10      # 220 = 20 rounds over the eleven task shapes


@pytest.mark.parametrize("part", range(FUZZ_SLICES))
def test_fuzz_parity_slice(part):
    rng = np.random.default_rng(FUZZ_SEED)
    cases = [_fuzz.draw_case(rng, i) for i in range(FUZZ_CASES)]          # the whole sweep is drawn, one slice is run
    per = FUZZ_CASES // FUZZ_SLICES
    seen = set()
    for c in cases[part * per:(part + 1) * per]:
        seen.add(_fuzz.run_case(c))
    assert len(seen) >= 3                                                 # a slice crosses several kernel families


def _relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


@pytest.mark.parametrize("lam", [1e-4, 1e-2, 10.0])
@pytest.mark.parametrize("task,T,kw", [("panda_reaching", 3000, {}), ("panda_reaching", 1500, dict(dense_residuals=True)),
                                       ("acrobot", 800, dict(config_id=1, dense_residuals=True))])
def test_running_inverse_stress(task, T, kw, lam):
    """tools/ns_stress.py: the Newton-Schulz running inverse of Quu + lambda I over long horizons at the ends of the
    lambda range [1e-4, 10] (Optimiser.h:239-242), fused and materialising, K and k against the oracle."""
    p = synth.make_problem(task=task, T=T, batch=2, min_N=5, **kw)
    ref = [pipeline.run_trajectory(p, b, lam=lam, stages=("fd", "interp", "cost", "bwd")) for b in range(2)]
    for fused in (True, False):
        with Engine(p["dof"], p["m"], T, p["nr"], batch=2, fused=fused) as e:
            synth.upload(e, p)
            e.fd_difference()
            if not fused:
                e.interpolate(); e.cost_derivs()
            st, dJ = e.backward(lam, 100)
            K, k = e.gains()
        for b, o in enumerate(ref):
            assert o["status"] == st[b]
            if st[b] == 0:
                assert _relerr(K[b], o["K"]) < 1e-9 and _relerr(k[b], o["k"]) < 1e-9, (fused, b)
