#!/bin/bash
# Builds the host emulation of the device headers (tests/host_emul/hostemu.cpp) with UBSan and then with ASan and runs the CPU suites
# that drive it.  The device headers are plain C++ shared between hipcc and g++, so undefined shifts, signed overflow and out-of-bounds
# indexing in field/scalar/group/ring code show up here without a GPU.  The regular library is put back afterwards.
set -e
cd "$(dirname "$0")/../.."
EMU=tests/host_emul/libs2k_hostemu.so
KEEP=$(mktemp); cp "$EMU" "$KEEP"; trap 'cp "$KEEP" "$EMU"; rm -f "$KEEP"' EXIT
SUITES="tests/test_cpu_oracle.py tests/test_cpu_adversarial.py"
echo "== UBSan"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=undefined -fno-sanitize-recover=undefined -o "$EMU" tests/host_emul/hostemu.cpp
timeout 1500 python -m pytest $SUITES -x -q -p no:cacheprovider
echo "== ASan"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address -o "$EMU" tests/host_emul/hostemu.cpp
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) timeout 1500 python -m pytest $SUITES -x -q -p no:cacheprovider
