"""Randomised parity sweep on the GPU box: python tools/fuzz_parity.py [cases] [seed].  The cases are those of
tests/_fuzz.py; a fixed-seed slice of the same sweep runs in the -m gpu suite (tests/test_gpu_fuzz.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(".", "tests"))
import _fuzz

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst = {}
t0 = time.time()
for case in range(N):
    c = _fuzz.draw_case(rng, case)
    var = _fuzz.run_case(c, worst)
    print(f"case {case:3d} {c['task']:15s} T={c['T']:5d} B={c['batch']} min_N={-1 if c['ragged'] else c['min_N']} lam={c['lam']:.2e} pd={c['pd']:3d} {var}  ok", flush=True)
print(f"{N} cases in {time.time() - t0:.1f} s; worst relative errors per kernel family:")
for var, w in sorted(worst.items()):
    print(f"  {var:55s} " + " ".join(f"{k}={v:.1e}" for k, v in w.items()))
