#!/bin/bash
# CPU-only sanitizer runs of the C++ host layer (scheduler, channels, CBOR codec, node protocol loopback):
# AddressSanitizer+UBSan, then ThreadSanitizer.  GPU sanitizers are not available on this pool.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO/flux_amd/host
SRC="flux_host_test.cpp flux_host.cpp yaml_lite.cpp cbor.cpp flux_net.cpp"
LINK="-L.. -lflux_hip -Wl,-rpath,$REPO/flux_amd -Wl,-rpath,/opt/rocm/lib"
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/flux_host_test_asan $SRC $LINK
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o /tmp/flux_host_test_tsan $SRC $LINK
cd $REPO
ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0 /tmp/flux_host_test_asan scenes /tmp | tail -1
/tmp/flux_host_test_tsan scenes /tmp | tail -1
# the threaded BVH builder (csrc/bvh.cpp, round 6): its self-test -- the 120 000-triangle threaded build against the serial one -- under both
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/bvh_selftest_asan tests/bvh_selftest.cpp flux_amd/csrc/bvh.cpp
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o /tmp/bvh_selftest_tsan tests/bvh_selftest.cpp flux_amd/csrc/bvh.cpp
FLUX_BUILD_THREADS=4 ASAN_OPTIONS=detect_leaks=1 /tmp/bvh_selftest_asan | tail -1
FLUX_BUILD_THREADS=4 /tmp/bvh_selftest_tsan | tail -1
