# timeline of the workflow leg's last repetition: kernel, start (ms from the first kernel of the repetition), duration, gap to the previous kernel's end
export TMPDIR=/tmp; mkdir -p gpurun_out/wftl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wftl -o wf -- python bench.py --steps 1 --warmup 0 --stage-inputs 0 --strict-steps 0 --config1-steps 0 --cpu-samples 0 --fit-concordance 0 --verify-columns 0 --kernel-alone 0 --regimes 0 --dropin 0 --workflow-reps 2 > gpurun_out/wftl/log 2>&1
python - <<'PY' > gpurun_out/wf_timeline.txt
import csv
rows = list(csv.DictReader(open('gpurun_out/wftl/wf_kernel_trace.csv')))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')) for r in rows))
# the last k_rs_rowtotal starts the last repetition's reference sets (one per cohort)
idx = [i for i, e in enumerate(ev) if 'k_link_copy' in e[2] or 'k_rs_rowtotal' in e[2]]
first = [i for i, e in enumerate(ev) if 'k_rs_rowtotal' in e[2]][-1]
# go back to the upload's first kernel
i0 = first
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 3_000_000 and (first - i0) < 400: i0 -= 1
t0 = ev[i0][0]
prev_end = t0
busy = 0
for s, e, n in ev[i0:]:
    print("%9.3f  dur %8.3f  gap %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, n))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.3f ms, kernel sum %.3f ms" % ((prev_end - t0) / 1e6, busy / 1e6))
PY
rm -rf gpurun_out/wftl
