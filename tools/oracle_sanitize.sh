#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow: the class of bug that made sin(1e38) differ between
# the two sides in round 5).  CPU only -- GPU sanitizers are not available on the pool.  The oracle's own tests, the shared libm's and the texture
# tests run against the instrumented build; any report ends the run (halt_on_error).  Usage: tools/oracle_sanitize.sh [pytest args]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/liboracle_san.so
g++ -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -fopenmp -Wall -Wno-unused-function -Wno-unknown-pragmas \
    -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -march=native -shared \
    -o "$OUT" "$ROOT/oracle/oracle.cpp"
cd "$ROOT"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ORK_LIB="$OUT" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_oracle_golden.py tests/test_oracle_bsdf.py tests/test_oracle_intersect.py tests/test_oracle_render.py tests/test_libm.py tests/test_textures.py -x -q "$@"
