#!/bin/bash
# The closing sweeps of a round on the final library, back to back (run through gpurun from the repo root):  tools/fuzz_campaign.sh r06 [scale]
# scale multiplies the seconds of every sweep (1 = 27 minutes).
TAG=${1:-rNN}; K=${2:-1}
OUT=gpurun_out/${TAG}_fuzz_final.txt
s() { python -c "print(int($1 * $K))"; }
echo "Closing sweeps of round ${TAG#r} on the final library (kernel sources $(python -c 'from exomedepth_amd import _build; print(_build.csrc_sha16())')), one MI355X box:" > $OUT
for t in "fuzz_tables.py $(s 420) 601" "fuzz_cohort.py $(s 300) 602" "fuzz_parity.py $(s 240) 603" "fuzz_more.py $(s 180) 604" "fuzz_fit_sm.py $(s 150) 605" "fuzz_bins.py $(s 120) 606" "fuzz_refcohort.py $(s 240) 607"; do
  echo "--- tools/$t" >> $OUT
  python tools/$t 2>&1 | tail -8 >> $OUT
done
cat $OUT
