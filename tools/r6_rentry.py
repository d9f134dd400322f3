import sys, time, json, os
sys.path.insert(0,'.')
import numpy as np, torch
torch.cuda.init()
import exomedepth_amd as ed
from exomedepth_amd import synth
import bench
E,S=200000,1024
chrom_off,start,end=synth.exon_design(E,24,seed=20250620)
dev=torch.device('cuda',0)
test,ref,p,phi=synth.counts_torch(chrom_off,S,dev,seed=20250623)
print(os.cpu_count(), len(os.sched_getaffinity(0)))
for rep in range(2):
    print(json.dumps(bench.r_entry_leg(ed,chrom_off,start,end,test,ref,S,4,1,phi,p)))
