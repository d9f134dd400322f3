"""Cohorts back to back: when does what run?  The next cohort's upload (a host thread, a copy stream) against this cohort's reference-set stage and calls;
prints per upload (start, copy done, widened) and per cohort (start, reference sets done, calls done) in ms.  With ED_REFCOHORT_TIMING=1 the library
prints its phases: how the stage's small host <-> device copies were found waiting behind the 410-MB upload (round 5; csrc/edrefcohort.inc::RcLink).
NCH=<n>: the upload cut into n copies (changes nothing: the engine takes them in order).    python tools/b2b_probe.py"""
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S, C = 200_000, 1024, 24
dev = torch.device("cuda:0")
chrom_off, start, end = synth.exon_design(E, C, 1)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, 1)
plan = ed.Plan(chrom_off, start, end)
pin = ed.PinnedArray((E, S), np.uint16); pin.array[...] = test.cpu().numpy()
bl = (np.asarray(end) - np.asarray(start)) / 1000.0
co = ed.Cohort(plan, S, 1, emit_mode=2, counts_layout=1)
ref_t = torch.empty((S, E), dtype=torch.int32, device=dev); counts_sm = torch.empty((S, E), dtype=torch.int32, device=dev)
cs, ws = torch.cuda.Stream(), torch.cuda.Stream()
T0 = time.perf_counter()
log = []
class Upload(threading.Thread):
    def run(self):
        a = time.perf_counter()
        torch.cuda.set_device(dev)
        with torch.cuda.stream(cs):
            NCH = int(os.environ.get("NCH", "1"))
            raw = torch.empty((E, S), dtype=torch.int16, device=dev)
            src = torch.from_numpy(pin.array.view(np.int16))
            for q in range(NCH):
                r0, r1 = E * q // NCH, E * (q + 1) // NCH
                raw[r0:r1].copy_(src[r0:r1], non_blocking=True)
            cs.synchronize(); b = time.perf_counter()
            self.d = raw.view(torch.int16).to(torch.int32) & 0xffff
            cs.synchronize()
        log.append(("upload", a - T0, b - T0, time.perf_counter() - T0))
nxt = Upload(); nxt.start(); nxt.join()
for k in range(5):
    nxt.join(); d = nxt.d
    a = time.perf_counter()
    nxt = Upload(); nxt.start()
    rs = ed.cohort_select_reference_sets(d, bl, 10000, max_refs=32, reference_out=ref_t, sample_major=True, counts_sm_out=counts_sm, stream=ws.cuda_stream)
    b = time.perf_counter()
    tk = co.submit(counts_sm, ref_t, n_samples=S, ready_stream=ws.cuda_stream); co.wait(tk)
    c = time.perf_counter()
    log.append(("iter", a - T0, b - T0, c - T0))
nxt.join()
for l in sorted(log, key=lambda x: x[1]): print(l[0], " ".join("%.2f" % (1e3 * v) for v in l[1:]))
