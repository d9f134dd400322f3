#!/bin/bash
export TMPDIR=/tmp
cat > /tmp/ft.py <<'PY'
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
E, S = 200_000, 1024
dev = torch.device("cuda:0")
chrom_off, start, end = synth.exon_design(E, 24, 20250623)
plan = ed.Plan(chrom_off, start, end)
test, ref, p, phi = synth.counts_torch(chrom_off, S, dev, seed=7, mean_depth=100.0)
t_in, r_in = test.t().contiguous(), ref.t().contiguous()
batch = ed.Batch(plan, S); batch.set_emit_mode(2); batch.set_counts_layout(1)
dphi = torch.zeros(S, dtype=torch.float64, device=dev); dexp = torch.zeros(S, dtype=torch.float64, device=dev)
for _ in range(6):
    batch.fit(t_in, r_in, dphi, dexp)
torch.cuda.synchronize()
print("unconverged", batch.fit_unconverged())
PY
for M in 1 2 4 8 100; do
  rm -rf /tmp/ftp; mkdir -p /tmp/ftp
  ED_FIT_MAXIT=$M timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ftp -o ft -- python /tmp/ft.py > /tmp/ftp/log 2>&1
  grep unconverged /tmp/ftp/log
  python - $M <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/ftp/**/ft_kernel_stats.csv', recursive=True)[0]
for r in csv.reader(open(f)):
    if 'hg8::k_fit_hnewton' in r[0]: print('maxit', sys.argv[1], 'hnewton avg_ms', float(r[3])/1e6, 'min', float(r[5])/1e6, 'max', float(r[6])/1e6)
PY
done
