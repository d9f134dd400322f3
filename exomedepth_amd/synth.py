"""Synthetic exon designs and count matrices.

Not part of the hot path: these generators only feed the parity tests (numpy, small sizes) and
bench.py (torch on the GPU, full sizes).  The model is the caller's own: reference counts are
Poisson around K times the exon depth, test counts are beta-binomial given the total with the
per-state proportion p' = p*c/(p*c + 1 - p) of reference src/CNV_estimate.cpp:75-77, c in {0.5, 1, 1.5}.
"""
import numpy as np

# exons per chromosome proportional to the real autosome+X+Y exon counts (approximate hg19 exome)
_CHROM_WEIGHTS = np.array([19.0, 13.5, 11.2, 7.7, 8.7, 10.1, 9.3, 6.9, 7.9, 7.8, 11.3, 10.5, 3.4, 6.4, 7.0, 8.7,
                           11.6, 2.9, 12.6, 5.2, 2.2, 4.4, 7.7, 0.5])


def exon_design(n_exons, n_chrom=24, seed=0):
    """Returns (chrom_off int32[n_chrom+1], start int32[n_exons], end int32[n_exons])."""
    rng = np.random.default_rng(seed)
    w = _CHROM_WEIGHTS[:n_chrom] if n_chrom <= 24 else np.ones(n_chrom)
    sizes = np.floor(w / w.sum() * n_exons).astype(np.int64)
    sizes[0] += n_exons - sizes.sum()
    chrom_off = np.zeros(n_chrom + 1, dtype=np.int32)
    chrom_off[1:] = np.cumsum(sizes)
    start = np.empty(n_exons, dtype=np.int64)
    gaps = np.clip(np.round(rng.lognormal(np.log(3000.0), 1.5, n_exons)), 50, 5e6).astype(np.int64)
    length = rng.integers(50, 501, n_exons)
    for c in range(n_chrom):
        lo, hi = chrom_off[c], chrom_off[c + 1]
        if hi > lo:
            pos = np.cumsum(gaps[lo:hi] + 500)
            # keep every chromosome inside int32 (the reference's positions are R integers)
            if pos[-1] > 2_000_000_000:
                pos = (pos * (2_000_000_000 / pos[-1])).astype(np.int64)
            start[lo:hi] = pos
    end = start + length
    return chrom_off, start.astype(np.int32), end.astype(np.int32)


def sample_params(n_samples, seed=0, K=8.0):
    """Per-sample (size factor, p, phi) as in SURVEY.md section 8d."""
    rng = np.random.default_rng(seed + 7919)
    sf = rng.lognormal(0.0, 0.25, n_samples)
    p = sf / (sf + K)
    phi = rng.uniform(0.002, 0.01, n_samples)
    return sf, p, phi


def counts_numpy(chrom_off, n_samples, seed=0, K=8.0, n_segments=40, mean_depth=100.0):
    """(test int32 [E][S], ref int32 [E][S], p[S], phi[S], copy_ratio int8 [E][S]) with planted CNVs."""
    rng = np.random.default_rng(seed)
    E = int(chrom_off[-1])
    sf, p, phi = sample_params(n_samples, seed, K)
    lam = rng.lognormal(np.log(mean_depth), 0.8, E)
    state = np.zeros((E, n_samples), dtype=np.int8)  # 0 normal, 1 deletion, 2 duplication
    for s in range(n_samples):
        for _ in range(n_segments):
            ln = rng.geometric(1.0 / 6.0)
            st = rng.integers(0, max(E - ln, 1))
            state[st:st + ln, s] = 1 if rng.random() < 2.0 / 3.0 else 2
    cr = np.array([1.0, 0.5, 1.5])[state]
    ref = rng.poisson(K * lam[:, None], size=(E, n_samples)).astype(np.int64)
    tot = rng.poisson((sf[None, :] * cr + K) * lam[:, None]).astype(np.int64)
    tot = np.maximum(tot, 0)
    pp = p[None, :] * cr / (p[None, :] * cr + 1 - p[None, :])
    theta = (1 - phi) / phi
    a = pp * theta[None, :]
    b = (1 - pp) * theta[None, :]
    lamb = rng.beta(a, b)
    test = rng.binomial(tot, lamb)
    # the model conditions on the total; keep test + ref = that total so that `ref` is what a caller holds
    ref = tot - test
    return test.astype(np.int32), ref.astype(np.int32), p, phi, state


def counts_torch(chrom_off, n_samples, device, seed=0, K=8.0, n_segments=40, mean_depth=100.0, chunk=4096):
    """Same model generated on the GPU with torch (bench.py; E x S of order 1e8..1e9 cells).
    Returns (test int32 [E][S], ref int32 [E][S], p float64[S], phi float64[S]) as device tensors."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    E = int(chrom_off[-1])
    S = int(n_samples)
    sf_np, p_np, phi_np = sample_params(S, seed, K)
    sf = torch.tensor(sf_np, device=device, dtype=torch.float64)
    p = torch.tensor(p_np, device=device, dtype=torch.float64)
    phi = torch.tensor(phi_np, device=device, dtype=torch.float64)
    test = torch.empty((E, S), device=device, dtype=torch.int32)
    ref = torch.empty((E, S), device=device, dtype=torch.int32)
    # planted segments: a sparse state matrix built from segment starts/lengths
    state = torch.zeros((E, S), device=device, dtype=torch.int8)
    nseg = n_segments
    starts = torch.randint(0, max(E - 64, 1), (nseg, S), device=device, generator=g)
    lens = torch.clamp((torch.empty((nseg, S), device=device).exponential_(1.0 / 6.0, generator=g)).long() + 1, max=64)
    kinds = (torch.rand((nseg, S), device=device, generator=g) < 2.0 / 3.0)
    cols = torch.arange(S, device=device)
    for k in range(nseg):
        for d in range(int(lens[k].max().item())):
            m = lens[k] > d
            rows = (starts[k] + d)[m]
            state[rows, cols[m]] = torch.where(kinds[k][m], 1, 2).to(torch.int8)
    theta = (1 - phi) / phi
    lam_all = torch.empty(E, device=device, dtype=torch.float64).log_normal_(float(np.log(mean_depth)), 0.8, generator=g)
    crv = torch.tensor([1.0, 0.5, 1.5], device=device, dtype=torch.float64)
    for lo in range(0, E, chunk):
        hi = min(lo + chunk, E)
        lam = lam_all[lo:hi, None]
        cr = crv[state[lo:hi].long()]
        tot = torch.poisson((sf[None, :] * cr + K) * lam, generator=g)
        pp = p[None, :] * cr / (p[None, :] * cr + 1 - p[None, :])
        a = (pp * theta[None, :]).float()
        b = ((1 - pp) * theta[None, :]).float()
        ga = torch._standard_gamma(a)
        gb = torch._standard_gamma(b)
        lamb = (ga / (ga + gb)).double().clamp_(0.0, 1.0)
        t = torch.binomial(tot, lamb, generator=g)
        test[lo:hi] = t.to(torch.int32)
        ref[lo:hi] = (tot - t).to(torch.int32)
    return test, ref, p, phi
