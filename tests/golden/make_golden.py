#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Runs ONLY in the build container, where /root/reference
exists: it (a) calls the reference's own special-function sources, compiled as they lie into
oracle/_ref/libgslsf_ref.so by oracle/Makefile, and stores inputs + outputs; (b) converts the
reference's bundled data file data/ExomeCount.RData into a small numpy fixture; (c) stores the
checker's (libm flavour) outputs on that data as regression vectors.  No reference source text is
copied: fixtures are inputs and expected outputs only.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import edoracle as eo  # noqa: E402
import rdata_reader  # noqa: E402


def around(vals, rel=(1e-15, 1e-12, 1e-6, 1e-3)):
    out = []
    for v in vals:
        out.append(v)
        for r in rel:
            out += [v * (1 - r), v * (1 + r)]
        out += [np.nextafter(v, 0), np.nextafter(v, np.inf)]
    return np.array(out)


def negative_domain_fixture():
    """sf_ref_negative.npz: gsl_sf_lngamma_sgn_e of the reference build (value, sign) on negative non-integer arguments --
    reflection branch, next to -1, next to -N (lngamma_sgn_sing with its psi_n terms), far negative, and around 0."""
    rng = np.random.default_rng(20250929)
    x = np.concatenate([-rng.uniform(0.02, 60, 1500), -1 + rng.uniform(-0.0149, 0.0149, 300),
                        -rng.integers(2, 300, 700) + rng.uniform(-0.0149, 0.0149, 700),
                        -rng.integers(2, 400, 300) + rng.choice([1e-12, -1e-9, 3e-6, -2.5e-4, 0.0011, -0.0051, 0.0101], 300),
                        -rng.integers(170, 100000, 200) + rng.uniform(-0.0149, 0.0149, 200),
                        -np.exp(rng.uniform(np.log(1e3), np.log(1e9), 200)), rng.uniform(-0.02, 0.02, 100)])
    x = x[(x != np.floor(x))]
    v, s, st = eo.ref_lngamma_sgn(x)
    assert np.all(st == 0)
    np.savez_compressed(os.path.join(HERE, "sf_ref_negative.npz"), lngamma_sgn_x=x, lngamma_sgn=v, lngamma_sgn_sign=s)


def survey_probe_fixture():
    """config1_survey_probe.json: the facts SURVEY.md 8c (G4) recorded from the reference's compiled C for Exome1 vs
    Exome2+3+4 with (eta, phi) = (-1.36727, 0.0049568) -- those are `reference_facts`, typed in from the survey, not
    computed here -- and, beside them, the checker's call table for exactly those parameters (`calls_regression`)."""
    import math
    d = np.load(os.path.join(HERE, "exomecount_chr1.npz"))
    counts = d["counts"]
    test = counts[:, 0].astype(np.int32)
    ref = counts[:, 1:].sum(axis=1).astype(np.int32)
    facts = {"eta": -1.36727, "phi": 0.0049568, "padded_observations": 26549, "state_counts_padded": [26320, 121, 108], "n_calls": 25}
    p = 1.0 / (1.0 + math.exp(-facts["eta"]))
    L, _ = eo.get_loglike_matrix(facts["phi"], p, test + ref, test, 1.0, eo.LIBM)
    path, calls = eo.callcnvs(L, np.array([0, test.size], np.int32), d["start"], d["end"])
    json.dump({"reference_facts": facts, "calls_regression": calls.astype(np.int64).tolist()},
              open(os.path.join(HERE, "config1_survey_probe.json"), "w"), indent=1)


def main():
    eo.build()
    if "--survey-probe-only" in sys.argv:
        survey_probe_fixture()
        return
    if "--negative-only" in sys.argv:
        negative_domain_fixture()
        return
    assert eo.ref_available(), "oracle/_ref/libgslsf_ref.so missing (needs /root/reference)"
    rng = np.random.default_rng(20250620)
    g = {}
    # ---- G1: special functions of the reference build (error-free domain only) ----
    n = 3000
    x = np.exp(rng.uniform(np.log(1e-3), np.log(1e7), n))
    y = np.exp(rng.uniform(np.log(1e-3), np.log(1e7), n))
    base = np.exp(rng.uniform(-2, 12, 400))
    seam = np.concatenate([base * r for r in (0.2, np.nextafter(0.2, 0), np.nextafter(0.2, 1), 0.1999, 0.2001)])
    g["lnbeta_x"] = np.concatenate([x, seam, rng.uniform(0.5, 300, n) + rng.integers(0, 500, n)])
    g["lnbeta_y"] = np.concatenate([y, np.tile(base, 5), rng.uniform(5, 3000, n) + rng.integers(0, 3000, n)])
    g["lnbeta"] = eo.ref_call2("gsl_sf_lnbeta", g["lnbeta_x"], g["lnbeta_y"])
    g["lngamma_x"] = np.concatenate([np.exp(rng.uniform(np.log(1e-6), np.log(1e9), n)), around([0.02, 0.5, 0.99, 1.0, 1.01, 1.99, 2.0, 2.01]),
                                     rng.uniform(0.02, 0.5, 300)])
    g["lngamma"] = eo.ref_call1("gsl_sf_lngamma", g["lngamma_x"])
    g["gammastar_x"] = np.concatenate([np.exp(rng.uniform(np.log(1e-4), np.log(1e17), n)), around([0.5, 2.0, 10.0, 8192.0, 1 / 2.2204460492503131e-16]),
                                       rng.uniform(0.02, 0.5, 300)])
    g["gammastar"] = eo.ref_call1("gsl_sf_gammastar", g["gammastar_x"])
    g["log_1plusx_x"] = np.concatenate([rng.uniform(-0.99, 3, n), rng.uniform(-3e-3, 3e-3, n), around([2.4607833005759251e-03, 0.5, -0.5, 0.2])])
    g["log_1plusx"] = eo.ref_call1("gsl_sf_log_1plusx", g["log_1plusx_x"])
    g["psi_x"] = np.exp(rng.uniform(np.log(1e-2), np.log(1e7), n))
    g["psi"] = eo.ref_call1("gsl_sf_psi", g["psi_x"])
    g["psi_1"] = eo.ref_call1("gsl_sf_psi_1", g["psi_x"])
    np.savez_compressed(os.path.join(HERE, "sf_ref.npz"), **g)

    # ---- bundled data -> fixture ----
    d = rdata_reader.exome_count("/root/reference/data/ExomeCount.RData")
    start = np.array(d["start"], dtype=np.int32)
    end = (start + np.array(d["width"], dtype=np.int32) - 1).astype(np.int32)
    counts = np.stack([np.array(d["Exome%d" % i], dtype=np.float64) for i in range(1, 5)], axis=1)
    assert np.all(counts == np.round(counts))
    np.savez_compressed(os.path.join(HERE, "exomecount_chr1.npz"), start=start, end=end, counts=counts.astype(np.int32))

    # ---- G4: config 1 through the checker (libm flavour) ----
    E = start.size
    chrom_off = np.array([0, E], dtype=np.int32)
    exp = {}
    summary = {}
    for i in range(4):
        test = counts[:, i].astype(np.int32)
        ref = (counts.sum(axis=1) - counts[:, i]).astype(np.int32)
        phi, p, ll, it = eo.fit_mle(test, ref)
        L, nerr = eo.get_loglike_matrix(phi, p, test + ref, test, 1.0, eo.LIBM)
        path, calls = eo.callcnvs(L, chrom_off, start, end)
        exp["phi%d" % i] = phi; exp["p%d" % i] = p
        exp["path%d" % i] = path
        exp["calls%d" % i] = calls
        rows = np.arange(0, E, 97)
        exp["ll_rows%d" % i] = L[rows]
        summary["sample%d" % (i + 1)] = {"phi": phi, "p": p, "gsl_errors": int(nerr), "ncalls": int(len(calls)),
                                         "state_counts": np.bincount(path, minlength=3).tolist(),
                                         "loglik_sha256": hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest()}
    np.savez_compressed(os.path.join(HERE, "config1_expected.npz"), **exp)
    json.dump(summary, open(os.path.join(HERE, "config1_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
    survey_probe_fixture()
    negative_domain_fixture()


if __name__ == "__main__":
    main()
