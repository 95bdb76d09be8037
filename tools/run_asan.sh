#!/bin/bash
# The host side of libdce.so under AddressSanitizer + UBSan (python -m deep_contact_estimator_amd.build --asan: host objects
# instrumented, device code unchanged): the ABI tests, the host-logic tests, two contexts on two threads, and the two plain-C
# clients.  Run on the GPU box:  tools/run_asan.sh [out-file]     (default gpurun_out/r4_asan.txt)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/r4_asan.txt}; mkdir -p "$(dirname "$OUT")"
LIB=deep_contact_estimator_amd/libdce_asan.so
[ -f $LIB ] || python -m deep_contact_estimator_amd.build --asan >/dev/null || exit 1
RT=$(python -c "from deep_contact_estimator_amd import build; print(build.asan_runtime())")
CLANG=$(dirname "$(python -c "from deep_contact_estimator_amd import build; print(build._hipcc())")")/../lib/llvm/bin/clang
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  echo "== $(date -u +%FT%TZ) libdce_asan.so ($(python -c "from deep_contact_estimator_amd import build; print(build.source_hash()[:12])")), runtime $RT"
  echo "== pytest: ABI + host logic + two contexts on two threads (LD_PRELOAD of the sanitizer runtime, DCE_LIB=$LIB)"
  LD_PRELOAD=$RT DCE_LIB=$PWD/$LIB python -m pytest -q -p no:cacheprovider tests/test_abi.py tests/test_round3_cpu.py \
      "tests/test_gpu_parity.py::test_two_contexts_on_two_threads" "tests/test_gpu_parity.py::test_error_behaviour" 2>&1 | tail -15
  echo "== plain C clients built with -fsanitize=address,undefined against the instrumented library"
  TMP=$(mktemp -d); ln -sf $PWD/$LIB $TMP/libdce.so
  $CLANG -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -fopenmp=libgomp -Iinclude -Ioracle tests/c/abi_client.c oracle/dce_oracle.c \
      -L$TMP -ldce -L/opt/rocm/lib -lamdhip64 -lm -lgomp -Wl,-rpath,$TMP -Wl,-rpath,/opt/rocm/lib -shared-libasan -o $TMP/abi_client 2>&1 | tail -5
  LD_LIBRARY_PATH=$(dirname $RT):${LD_LIBRARY_PATH:-} $TMP/abi_client 2>&1 | tail -6
  $CLANG -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -Iinclude tests/c/abi_ranks.c -L$TMP -ldce -L/opt/rocm/lib -lamdhip64 -lm \
      -Wl,-rpath,$TMP -Wl,-rpath,/opt/rocm/lib -shared-libasan -o $TMP/abi_ranks 2>&1 | tail -5
  LD_LIBRARY_PATH=$(dirname $RT):${LD_LIBRARY_PATH:-} DCE_COMM_ID_FILE=$TMP/id tools/launch_ranks.sh 1 $TMP/abi_ranks --windows 300 2>&1 | grep -v "^    #[0-9]* 0x.*sanitizer\|longer_pathname" | tail -40
  echo "== done: any 'ERROR: AddressSanitizer' / 'runtime error:' line above is a finding.  (Known noise, not one: 'AddressSanitizer: CHECK failed:"
  echo "   sanitizer_allocator_device.h ... dev_runtime_unloaded_' AFTER a client's OK line -- ROCm's sanitizer runtime asserting inside libhsa-runtime64's own"
  echo "   exit-time teardown (__cxa_finalize -> libamdhip64 -> libhsa-runtime64 -> operator delete); no frame of libdce.so or of the client in it.)"
} > "$OUT" 2>&1
grep -c "ERROR: AddressSanitizer\|runtime error:" "$OUT"
tail -40 "$OUT"
