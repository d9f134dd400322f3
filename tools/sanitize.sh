#!/bin/bash
# Host side of the library under the sanitizers: libedcore.so's host code (streams, the stager's worker
# threads, the quantile selection threads, the mutex-guarded scratch of the cohort reference sets), shim/edcore_shim.c and the
# miniature R runtime of the tests, built with -fsanitize=address,undefined (and, second pass, -fsanitize=thread); device code is
# left alone (-fno-gpu-sanitize).  Python is not instrumented: the sanitizer runtime is preloaded.
#   tools/sanitize.sh cpu  [log]     the -m "not gpu" tests that load the library / the shim (runs without a GPU)
#   tools/sanitize.sh gpu  [log]     + tests/test_gpu_cohort.py, tests/test_gpu_refcohort.py, tests/test_shim.py -m gpu on the GPU box
#   tools/sanitize.sh tsan [log]     tests/test_gpu_cohort.py, test_gpu_refcohort.py, test_gpu_multidevice.py (one host thread per device) under ThreadSanitizer
set -u
MODE=${1:-cpu}; LOG=${2:-gpurun_out/sanitize_$MODE.log}
mkdir -p $(dirname $LOG)
# The sanitizer RUNTIME is gcc's stock one (libasan / libubsan / libtsan of the image's gcc 11): ROCm's clang ships an ASan runtime
# that intercepts hsa_amd_memory_pool_allocate for device-side instrumentation and aborts ("out of memory") in a process whose
# device code is not instrumented.  The instrumentation itself is clang's (hipcc) for libedcore and gcc's for the shim: same ABI (v8).
ASANRT=$(gcc -print-file-name=libasan.so); UBSANRT=$(gcc -print-file-name=libubsan.so); TSANRT=$(gcc -print-file-name=libtsan.so)
CLANG=gcc
export TMPDIR=/tmp
if [ "$MODE" = "tsan" ]; then
  # (ThreadSanitizer: gcc 11's runtime cannot lay out its shadow on the GPU box's kernel -- "unexpected memory mapping", with or
  #  without setarch -R; LLVM's own runtime re-executes itself into a layout it can handle and has no HSA interceptors)
  VAR=tsan; PRE=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.tsan-x86_64.so); SHIMF="-fsanitize=thread"; CLANG=/opt/rocm/lib/llvm/bin/clang
  export TSAN_OPTIONS="report_signal_unsafe=0:history_size=4:exitcode=0:suppressions=$(pwd)/tools/tsan.supp:log_path=$LOG.tsan"
else
  VAR=asan; PRE="$ASANRT $UBSANRT"; SHIMF="-fsanitize=address,undefined -g"
  export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:exitcode=66:halt_on_error=0:log_path=$LOG.asan"
  export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$LOG.ubsan"
fi
python -c "from exomedepth_amd import _build; print(_build.build_variant('$VAR'))" > $LOG 2>&1 || { tail -20 $LOG; exit 1; }
export ED_LIB_VARIANT=$VAR ED_SHIM_CC=$CLANG ED_SHIM_CFLAGS="$SHIMF"
case $MODE in
  cpu)  T="tests/test_abi.py tests/test_host_logic.py tests/test_shim.py"; M='not gpu' ;;
  gpu)  T="tests/test_abi.py tests/test_host_logic.py tests/test_shim.py tests/test_gpu_cohort.py tests/test_gpu_refcohort.py tests/test_gpu_tables.py tests/test_gpu_multidevice.py tests/test_gpu_dropin.py"; M='gpu or not gpu' ;;
  tsan) T="tests/test_gpu_cohort.py tests/test_gpu_refcohort.py tests/test_gpu_multidevice.py tests/test_gpu_dropin.py"; M='gpu' ;;
esac
echo "== $MODE: LD_PRELOAD=$PRE python -m pytest $T -m \"$M\"" >> $LOG
# (deselected: the two tests that generate its data with torch on the GPU -- torch's own HIP initialisation does not find the device
#  under the preloaded runtime; nothing of this library is involved)
RUN=""
LD_PRELOAD="$PRE" timeout 3000 $RUN python -m pytest $T -q -m "$M" -p no:cacheprovider \
  --deselect tests/test_gpu_refcohort.py::test_config4_geometry_every_sample_against_all_others_500k_x_2048 \
  --deselect tests/test_gpu_refcohort.py::test_one_rank_of_eight_200k_x_8192 \
  --deselect tests/test_gpu_cohort.py::test_counts_produced_on_the_cohorts_own_stream >> $LOG 2>&1
echo "pytest exit code $?" >> $LOG
for f in $LOG.asan.* $LOG.ubsan.* $LOG.tsan.*; do [ -f "$f" ] && { echo "== $f" >> $LOG; head -c 20000 "$f" >> $LOG; rm -f "$f"; }; done
echo "== sanitizer reports in the log:" >> $LOG
grep -c "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" $LOG >> $LOG
tail -15 $LOG
