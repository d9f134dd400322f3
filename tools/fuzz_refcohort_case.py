"""the case tools/fuzz_refcohort.py kept (largest difference in the mean between the two forms) against the checker's long-double MLE"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import refset_oracle as ro
z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_refcohort_case.npz")
counts, bl, nred, t, i = z["counts"], z["bl"], int(z["nred"]), int(z["t"]), int(z["i"])
others = np.delete(np.arange(counts.shape[1]), t)
one = ro.select_reference_set_lean(counts[:, t], np.ascontiguousarray(counts[:, others]), bl, nred, raw_prefixes=(i,))
raw = one["raw"][i]
print("E %d S %d test %d prefix %d: mean_p columns %.12g row-major %.12g MLE %.12g | phi columns %.6g row-major %.6g MLE %.6g" %
      (counts.shape[0], counts.shape[1], t, i, z["cols"][0], z["cols"][1], raw["mean_p"], z["phi"][0], z["phi"][1], raw["phi"]))
