#!/bin/bash
# tools/run_sanitizers.sh [OUT] -- the host side under AddressSanitizer + UndefinedBehaviorSanitizer, on a CPU box (NEVER on the GPU
# pool: GPU sanitizers are not available there and the host twin is not meant to drive a device):
#   1. build: every host translation unit of libcvsteer_hip.so (make -C cvsteer_amd/csrc san), the oracle (make -C oracle san),
#      the file readers' fuzz harness (tests/cpp/fuzz_readers.cpp);
#   2. `pytest -m "not gpu"` against the instrumented twins (CVSTEER_HIP_LIB / ORACLE_LIB, libasan preloaded into python);
#   3. 10 000 mutated PGM / .npy files through the batch driver's readers.
# Writes the log to OUT (default profiles/r06_asan_cpu.txt); exit status 0 = no sanitizer report anywhere.
# The reference's CI does the same with its gtest (.travis.yml:48-51: sanitize-address, sanitize-leak toolchains).
cd "$(dirname "$0")/.." || exit 1
out=${1:-profiles/r06_asan_cpu.txt}
asan=$(gcc -print-file-name=libasan.so)
{
  echo "== tools/run_sanitizers.sh, $(gcc --version | head -1); AddressSanitizer + UndefinedBehaviorSanitizer, host side only (CPU box)"
  make -C cvsteer_amd/csrc -s san 2>&1 | grep -E "error|warning: (unused|comparison)" ; echo "build libcvsteer_hip_san.so: rc ${PIPESTATUS[0]}"
  make -C oracle -s san 2>&1; echo "build oracle/_san: rc $?"
  g++ -std=c++11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude tests/cpp/fuzz_readers.cpp -o tests/cpp/fuzz_readers; echo "build fuzz_readers: rc $?"
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Icvsteer_amd/csrc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/host_logic_san.cpp \
      cvsteer_amd/csrc/cvs_handle.cpp cvsteer_amd/csrc/cvs_tune.cpp cvsteer_amd/csrc/cvs_state.cpp cvsteer_amd/csrc/cvs_taps.cpp -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib \
      -o tests/cpp/host_logic_san; echo "build host_logic_san: rc $?"
  echo "-- pytest -m 'not gpu' with the instrumented twins (leak check off: the interpreter itself never frees everything)"
  LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    CVSTEER_HIP_LIB=$PWD/tools/libcvsteer_hip_san.so ORACLE_LIB=$PWD/oracle/_san/liboracle_cvsteer.so \
    python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider -k "not telemetry_lookup and not campaign_log" 2>&1 | tail -15   # (that test initialises torch.cuda, which does not survive a preloaded libasan; none of this repo's code runs in it)
  rc_py=${PIPESTATUS[0]}
  echo "pytest: rc $rc_py"
  echo "-- file readers, 10 000 mutated files (truncated headers / rasters, maxval > 255, negative, zero and overflowing sizes, 2 GiB claims)"
  ./tests/cpp/fuzz_readers 10000 2>&1 | tail -5
  rc_fz=${PIPESTATUS[0]}
  echo "fuzz_readers: rc $rc_fz"
  echo "-- host logic without a device (argument checks, overlap rules against a byte model, CVS_OPTS parser, state layouts, taps), leak detection on"
  ASAN_OPTIONS=detect_leaks=1 ./tests/cpp/host_logic_san 2>&1 | grep -v "CVS_OPTS: unknown name" | grep "host_logic_san\|ERROR\|runtime error" | tail -5
  ASAN_OPTIONS=detect_leaks=1 ./tests/cpp/host_logic_san > /dev/null 2>&1
  rc_hl=$?
  echo "host_logic_san: rc $rc_hl"
  echo "-- the readers with leak detection on (LeakSanitizer)"
  ASAN_OPTIONS=detect_leaks=1 ./tests/cpp/fuzz_readers 2000 2>&1 | tail -2
  rc_lk=${PIPESTATUS[0]}
  echo "fuzz_readers with leak detection: rc $rc_lk"
  if [ "$rc_py" = 0 ] && [ "$rc_fz" = 0 ] && [ "$rc_lk" = 0 ] && [ "$rc_hl" = 0 ]; then echo "RESULT: clean"; else echo "RESULT: FINDINGS"; fi
} > "$out" 2>&1
grep -q "^RESULT: clean" "$out"
