#!/bin/bash
# CPU only: build oracle/oracle.c with AddressSanitizer + UBSan and run its test file against that
# build (GPU sanitizers are not available on the pool; the C oracle is the native CPU code of the repo).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -march=x86-64-v3 -fopenmp -fPIC -shared -ffp-contract=off -fsanitize=address,undefined \
    -fno-omit-frame-pointer -o /tmp/liboracle_asan.so "$R/oracle/oracle.c" -lm
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 - <<PY
import sys, ctypes as C
sys.path[:0] = ["$R", "$R/tests"]
from oracle import c_oracle as CO
CO._lib = C.CDLL("/tmp/liboracle_asan.so")
for f in ("oracle_von_mises_3d", "oracle_comfe_mises", "oracle_comfe_drucker_prager"):
    getattr(CO._lib, f).restype = C.c_longlong
import pytest
sys.exit(pytest.main(["-x", "-q", "$R/tests/test_oracle_c.py", "-p", "no:cacheprovider"]))
PY
