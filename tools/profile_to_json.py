"""profiles/<tag>_pmc_FETCH_SIZE.csv + <tag>_pmc_WRITE_SIZE.csv -> profiles/<tag>_pmc_traffic.json
(bytes per launch, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied)."""
import csv, json, sys
tag = sys.argv[1]
def load(name):
    d = {}
    for r in csv.DictReader(open("profiles/%s_pmc_%s.csv" % (tag, name))):
        d[r["kernel"]] = (float(r["mean_per_launch"]), int(r["launches"]))
    return d
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {}
for k in f:
    out[k] = {"FETCH_SIZE_KB_per_launch": f[k][0], "WRITE_SIZE_KB_per_launch": w.get(k, (0, 0))[0], "launches_profiled": f[k][1], "steps_profiled": 3,   # profile_round.sh: --steps 2 --warmup 1
             
              "hbm_bytes_per_launch": (2.0 * f[k][0] + w.get(k, (0, 0))[0]) * 1024.0,
              "correction": "FETCH_SIZE x2 (coalesced streams, gfx950); WRITE_SIZE as reported"}
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(out["k_emit_batch"]))
