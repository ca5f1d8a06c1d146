#!/bin/bash
# ThreadSanitizer and AddressSanitizer + UBSan runs of the host runtime (libmjr_host: physics thread, event thread and the
# service / step-request callers around one recursive mutex and a handful of atomics), on the CPU test harness backend:
#   tools/run_sanitizers.sh            -> profiles/r05_sanitizers.txt (OUT=... overrides)
# SURVEY.md §5 (the reference: mujoco_ros/cmake/Sanitizers.cmake:3-43, ENABLE_SANITIZER_{ADDRESS,THREAD,UNDEFINED_BEHAVIOR}).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-profiles/r05_sanitizers.txt}
TESTS="tests/test_host_env.py tests/test_host_services.py tests/test_host_sharded.py tests/test_host_sensors_plugin.py"
make -s -C mujoco_ros_pkgs_amd/host sanitizers || exit 1
make -s -C tests/host_harness || exit 1
LOGDIR=$(mktemp -d)
export MJB_PREBUILT=1
make -s -C oracle all
{
echo "# host runtime under sanitizers ($(gcc --version | head -1)); tests: $TESTS -m 'not gpu'"
echo "# (first run, round 3: TSan reported two data races -- MujocoEnv::model_valid_ read by physicsLoop() outside the mutex while"
echo "#  loadWithModelAndData() writes it, and num_steps_until_exit_ written by stepBurst() while getPendingSteps() reads it from the"
echo "#  caller's thread; both are std::atomic now.  The reference has the same two accesses on plain members.)"
echo
echo "## ThreadSanitizer (libmjr_host_tsan.so)"
TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 log_path=$LOGDIR/tsan exitcode=0" \
  LD_PRELOAD=$(gcc -print-file-name=libtsan.so) MJR_HOST_LIBRARY=mujoco_ros_pkgs_amd/host/libmjr_host_tsan.so \
  python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
n=$(cat $LOGDIR/tsan.* 2>/dev/null | grep -c "WARNING: ThreadSanitizer")
echo "ThreadSanitizer reports: $n"
cat $LOGDIR/tsan.* 2>/dev/null | grep -A14 "WARNING: ThreadSanitizer" | head -120
echo
echo "## AddressSanitizer + UndefinedBehaviorSanitizer (libmjr_host_asan.so)"
ASAN_OPTIONS="detect_leaks=0 halt_on_error=0 log_path=$LOGDIR/asan exitcode=0" UBSAN_OPTIONS="print_stacktrace=1 log_path=$LOGDIR/ubsan" \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" MJR_HOST_LIBRARY=mujoco_ros_pkgs_amd/host/libmjr_host_asan.so \
  python -m pytest $TESTS -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
n=$(cat $LOGDIR/asan.* 2>/dev/null | grep -c "ERROR: AddressSanitizer")
u=$(cat $LOGDIR/ubsan.* 2>/dev/null | grep -c "runtime error")
echo "AddressSanitizer reports: $n   UBSan reports: $u"
cat $LOGDIR/asan.* $LOGDIR/ubsan.* 2>/dev/null | grep -B2 -A12 "ERROR: AddressSanitizer\|runtime error" | head -80
} > $OUT 2>&1
rm -rf $LOGDIR
cat $OUT
