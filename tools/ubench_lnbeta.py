"""Micro-benchmark of the device lnbeta by branch mix (run under rocprofv3 --kernel-trace; the k_eval_sf
durations, in call order, are the measurements).  Cases: ratio branch only, Lanczos branch only, 50/50
and 90/10 per-lane mixes, ratio branch with small arguments (Chebyshev Gamma*)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from exomedepth_amd._lib import lib, check

def run(x, y):
    out = np.empty_like(x)
    check(lib().ed_eval_sf(0, x.size, C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(out.ctypes.data)))
    return out

rng = np.random.default_rng(0)
n = 1 << 24
x = rng.uniform(50, 500, n)
cases = {
    "ratio_only": (x, x * rng.uniform(6, 12, n)),
    "lanczos_only": (x, x * rng.uniform(1, 4, n)),
    "mix_50_50": (x, x * np.where(rng.random(n) < 0.5, 8.0, 2.0)),
    "mix_90_10": (x, x * np.where(rng.random(n) < 0.9, 8.0, 2.0)),
    "ratio_small_args": (rng.uniform(0.6, 9, n), rng.uniform(60, 900, n)),
}
for k, (a, b) in cases.items():
    r = run(np.ascontiguousarray(a), np.ascontiguousarray(b))
    print(k, float(np.nansum(r)))
