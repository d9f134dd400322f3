"""What clock and power does the chip sustain under the emission kernel?  Emission launches back to back (given phi, nothing else
queued) for a few seconds while rocm-smi is sampled from a side thread; then the same for an idle GPU.
    python tools/clock_power_probe.py [seconds]"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
E, S = 200_000, 1024
dev = torch.device("cuda", 0)
chrom_off, start, end = synth.exon_design(E, 24, seed=20250620)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=20250620 + 3)
plan = ed.Plan(chrom_off, start, end)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            c = d.get("card0", {})
            sclk = next((v for k, v in c.items() if "sclk" in k.lower()), None)
            pw = next((v for k, v in c.items() if "power" in k.lower() and "W" in k), None)
            m = re.search(r"(\d+)\s*Mhz", str(sclk), flags=re.I)
            out.append((time.time(), int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:  # noqa: BLE001
            out.append((time.time(), None, None, repr(e)[:80]))
        time.sleep(0.15)


res = {}
for label in ("idle", "emissions"):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    if label == "emissions":
        co = ed.Cohort(plan, S, 1, timing=1)     # one slab in flight, given parameters: emission launches + their chains only
        while time.time() - t0 < secs:
            for _ in range(20):
                co.submit(test, ref, phi=phi, expected=p, n_samples=S)
            co.drain(); n += 20
        tot, nr, nf = co.stage_ms_total()
        res["emission_ms_per_slab"] = tot["emissions"] / max(nr, 1)
        co.close()
    else:
        time.sleep(min(secs, 2.0))
    stop.set(); th.join()
    clk = [o[1] for o in out if len(o) == 3 and o[1]]
    pw = [o[2] for o in out if len(o) == 3 and o[2]]
    res[label] = {"samples": len(out), "sclk_mhz_mean": sum(clk) / len(clk) if clk else None, "sclk_mhz_min": min(clk) if clk else None,
                  "sclk_mhz_max": max(clk) if clk else None, "power_w_mean": sum(pw) / len(pw) if pw else None,
                  "power_w_max": max(pw) if pw else None, "first_raw": out[:1]}
r = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True)
res["max_power_raw"] = r.stdout.strip()[:300]
print(json.dumps(res))
