#!/bin/bash
# Sanitizer run of the CPU oracle (GPU sanitizers are not available on the pool; this is the CPU build).
set -e
cd "$(dirname "$0")/.."
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -o /tmp/libhmj_oracle_asan.so oracle/hmj_oracle.c -lm -lpthread
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 tools/oracle_asan_check.py
