#!/bin/bash
# build the library and fail loudly (header syntax as C, every .hip compiled, the .so linked)
set -e
cd "$(dirname "$0")/../.."
gcc -fsyntax-only -x c include/boficap_hip.h
python -m boficap_amd.build > /tmp/bofi_build.log 2>&1 || { grep -E "error" -A3 /tmp/bofi_build.log | head -40; echo "BUILD FAILED"; exit 1; }
tail -1 /tmp/bofi_build.log
