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
