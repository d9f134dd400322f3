# round 6's records that are not part of profile_round.sh (run through gpurun from the repo root; copies go to profiles/ by hand)
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --driver multi-device --gpus 1 --steps 5 --warmup 2 > gpurun_out/r06_multi_n1.json 2>> gpurun_out/r06_bench_default.err
python bench.py --driver multi-device --gpus 2 --devices 0,0 --steps 5 --warmup 2 > gpurun_out/r06_multi_n2_same_gpu.json 2>> gpurun_out/r06_bench_default.err
if [ "${1:-}" = "all" ]; then
python tools/concordance.py --samples 1024 --emit-mode tables --margins > gpurun_out/r06_concordance_tables_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
python tools/concordance.py --samples 1024 --emit-mode tables --margins --deep > gpurun_out/r06_concordance_tables_deep_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
python tools/concordance.py --samples 1024 --emit-mode tables --margins --depth 400 > gpurun_out/r06_concordance_tables_depth400_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
python tools/concordance.py --samples 1024 --emit-mode tables --margins --depth 1600 > gpurun_out/r06_concordance_tables_depth1600_200k_x_1024.json 2>> gpurun_out/r06_bench_default.err
fi
bash tools/sanitize.sh gpu gpurun_out/r06_sanitize_asan_ubsan_gpu.log > /dev/null 2>&1
bash tools/sanitize.sh tsan gpurun_out/r06_sanitize_tsan.log > /dev/null 2>&1
tail -4 gpurun_out/r06_sanitize_asan_ubsan_gpu.log; tail -4 gpurun_out/r06_sanitize_tsan.log
python - <<'PY'
import json
for f in ("gpurun_out/r06_multi_n1.json","gpurun_out/r06_multi_n2_same_gpu.json"):
    j=json.load(open(f)); print(f, j["ms_per_step"], j["devices"])
j=json.load(open("gpurun_out/r06_bench_default.json")); print(j["ms_per_step"], j["roofline"]["frac"], j["h2d"].get("r_entry"), j["h2d"]["pageable"], {k: v for k, v in j["extra"]["workflow"].items() if k.endswith("_ms") or k in ("choice_checksum_rank0", "reference_sets_form")})
PY
