#!/bin/bash
# Builds libsecp256k1 WITH the MI355X batch-verification hook compiled in, from a secp256k1-zkp source tree that is left untouched:
# the tree is copied, integration/secp256k1_amd_hook.patch is applied to the copy (three hunks: a CMake option, the include of the hook at the
# end of src/secp256k1.c, the redirect of the modules' secp256k1_ecmult_multi_var call sites), and the copy is built with CMake.
#   integration/build_hooked.sh <secp256k1-zkp source dir> <output dir> [extra cmake args]
# Result: <output dir>/build/lib/libsecp256k1.so* exporting secp256k1_amd_set_backend and the secp256k1_amd_* batch adapters next to the
# library's own API.  A maintainer applies the same patch in-tree and passes -DSECP256K1_ENABLE_AMD_HOOK=ON -DSECP256K1_AMD_HOOK_DIR=<this dir>.
set -euo pipefail
SRC=$(realpath "$1"); OUT=$(realpath -m "$2"); shift 2
HERE=$(cd "$(dirname "$0")" && pwd)
rm -rf "$OUT/src" "$OUT/build"; mkdir -p "$OUT"
cp -r "$SRC" "$OUT/src"
( cd "$OUT/src" && patch -p1 --no-backup-if-mismatch < "$HERE/secp256k1_amd_hook.patch" )
GEN=(); command -v ninja >/dev/null 2>&1 && GEN=(-G Ninja)
cmake -S "$OUT/src" -B "$OUT/build" "${GEN[@]}" -DCMAKE_BUILD_TYPE=Release -DSECP256K1_ENABLE_AMD_HOOK=ON -DSECP256K1_AMD_HOOK_DIR="$HERE" \
      -DSECP256K1_BUILD_TESTS=OFF -DSECP256K1_BUILD_EXHAUSTIVE_TESTS=OFF -DSECP256K1_BUILD_BENCHMARK=OFF -DSECP256K1_BUILD_EXAMPLES=OFF "$@" > "$OUT/cmake.log" 2>&1
cmake --build "$OUT/build" -j 8 > "$OUT/build.log" 2>&1
ls "$OUT"/build/lib/libsecp256k1.so*
