# the workflow leg alone (reference sets + calls), and the reference-set tests
python -m pytest tests/test_gpu_refcohort.py tests/test_gpu_refset.py tests/test_gpu_fit.py -q -m gpu -x 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --stage-inputs 0 --strict-steps 0 --config1-steps 0 --cpu-samples 0 --fit-concordance 0 --verify-columns 0 --kernel-alone 0 --regimes 0 --dropin 0 --workflow-reps 5 > gpurun_out/wf_line.json 2> gpurun_out/wf_err.txt
python - <<'PY'
import json
w=json.load(open("gpurun_out/wf_line.json"))["extra"]["workflow"]
print({k:w[k] for k in ("upload_ms","reference_sets_ms","calls_ms","total_ms","choice_checksum_rank0","references_chosen_mean","n_calls")}, w["back_to_back"]["ms_per_cohort"])
PY
