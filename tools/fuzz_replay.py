"""Find the parameters of one case of tools/fuzz_parity.py by replaying its random stream without doing the work:
    python tools/fuzz_replay.py <fuzz seed> <E> <S> <case seed>     -> gpurun_out/fuzz_case.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from exomedepth_amd import synth

rng = np.random.default_rng(int(sys.argv[1]))
want = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
for it in range(10_000_000):
    S = int(rng.choice([1, 3, 16, 63, 64, 65, 127, 512, 513, 520, 576, 640, 1000]))
    E = int(rng.integers(1, 40 if S > 400 else 600) * rng.choice([1, 7]))
    C = int(rng.integers(1, 6))
    seed = int(rng.integers(1 << 30))
    hit = seed == want[2]
    chrom_off, start, end = synth.exon_design(max(E, C), C, seed)
    E = int(chrom_off[-1])
    if rng.random() < 0.3 and C > 1:
        k = int(rng.integers(1, C))
        chrom_off = np.insert(chrom_off, k, chrom_off[k]).astype(np.int32)
        C += 1
    depth = float(rng.choice([3.0, 40.0, 150.0, 2500.0]))
    if hit:
        test, ref, p, phi, _ = synth.counts_numpy(chrom_off, S, seed, n_segments=2, mean_depth=depth)
    swap = rng.random() < 0.3
    if hit and swap:
        test, ref, p = ref.copy(), test.copy(), 1.0 - p
    if rng.random() < 0.5:
        dead = rng.random((E, S)) < 0.2
        if hit:
            test[dead] = 0; ref[dead] = 0
    mult = float(rng.choice([1.0, 1e-3, 30.0]))
    mixture = float(rng.choice([1.0, 1.0, 0.4]))
    tp = float(rng.choice([1e-4, 1e-2])); L = float(rng.choice([5e4, 2e3]))
    if hit:
        phi = np.minimum(phi * mult, 0.6)
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/fuzz_case.npz", chrom_off=chrom_off, start=start, end=end, test=test, ref=ref, p=p, phi=phi,
                            mixture=mixture, tp=tp, L=L)
        print("case", it, "E", E, "S", S, "C", C, "depth", depth, "swap", swap, "mult", mult, "mixture", mixture, "tp", tp, "L", L, "phi", phi[:3], "p", p[:3])
        break
    rng.choice(S, size=min(S, 6), replace=False)
    if E >= 200 and rng.random() < 0.5:
        rng.choice(S, size=min(S, 3), replace=False)
