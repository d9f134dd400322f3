"""The checker's fast MLE (sufficient statistics, oracle/edo_fit.inc::edo_fit_mle_hist) against its plain per-cell MLE."""
import numpy as np


def test_histogram_mle_equals_per_cell_mle(oracle):
    rng = np.random.default_rng(3)
    for case in range(8):
        E = int(rng.integers(300, 6000))
        depth = float(rng.choice([5.0, 60.0, 400.0, 3000.0]))
        lam = rng.lognormal(np.log(depth), 0.7, E)
        k = float(rng.uniform(1, 20))
        sig = float(rng.choice([0.0, 0.05, 0.2]))
        test = rng.poisson(lam).astype(np.int32)
        ref = rng.poisson(lam * k * rng.lognormal(0, sig, E) if sig else lam * k).astype(np.int32)
        if case % 3 == 0:
            test[::7] = 0; ref[::7] = 0
        a = oracle.fit_mle(test, ref)
        b = oracle.fit_mle_hist(test, ref)
        if a[0] > 1e-6:
            assert abs(a[1] - b[1]) <= 1e-11 * a[1] and abs(a[0] - b[0]) <= 1e-9 * a[0], (case, a, b)
        else:   # a (nearly) binomial column: the likelihood is flat in phi and lgamma(a + b ~ 1e9) carries ~1e-4 of noise
            assert b[0] < 1e-6 and abs(a[1] - b[1]) <= 1e-9 * a[1] and abs(a[2] - b[2]) <= 1e-8 * abs(a[2]), (case, a, b)
    assert oracle.fit_mle_hist(np.zeros(5, np.int32), np.zeros(5, np.int32))[3] == -1
