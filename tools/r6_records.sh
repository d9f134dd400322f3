# round 6's records that are not part of profile_round.sh
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --driver multi-device --gpus 1 --steps 5 --warmup 2 > gpurun_out/r06_multi_n1.json 2>> gpurun_out/r06_bench_default.err
python bench.py --driver multi-device --gpus 2 --devices 0,0 --steps 5 --warmup 2 > gpurun_out/r06_multi_n2_same_gpu.json 2>> gpurun_out/r06_bench_default.err
python tools/concordance.py --samples 1024 --emit-mode tables --margins --depth 400 > gpurun_out/r06_concordance_tables_depth400_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
python tools/concordance.py --samples 1024 --emit-mode tables --margins --depth 1600 > gpurun_out/r06_concordance_tables_depth1600_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
bash tools/sanitize.sh gpu gpurun_out/r06_sanitize_asan_ubsan_gpu.log > /dev/null 2>&1
bash tools/sanitize.sh tsan gpurun_out/r06_sanitize_tsan.log > /dev/null 2>&1
tail -4 gpurun_out/r06_sanitize_asan_ubsan_gpu.log; tail -4 gpurun_out/r06_sanitize_tsan.log
python - <<'PY'
import json
for f in ("gpurun_out/r06_concordance_tables_depth400_200k_x_1024.json","gpurun_out/r06_concordance_tables_depth1600_200k_x_1024.json"):
    j=json.load(open(f)); d=j["decision_margins"]
    print(f, j["tail_samples"], j["discordant_viterbi_states"], j["discordant_call_rows"], j["loglik_beyond_1e-10_relative"], j["max_relative_loglik_difference"], d["min_nonzero_margin"], d["max_abs_loglik_difference_device_vs_reference"], j["table_stats"])
for f in ("gpurun_out/r06_multi_n1.json","gpurun_out/r06_multi_n2_same_gpu.json"):
    j=json.load(open(f)); print(f, j["ms_per_step"], j["devices"])
j=json.load(open("gpurun_out/r06_bench_default.json")); print(j["ms_per_step"], j["h2d"].get("r_entry"), j["h2d"]["pageable"], j["extra"]["dropin"]["one_sample_sequence"]["next_sample"])
PY
