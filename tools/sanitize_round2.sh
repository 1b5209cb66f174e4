#!/bin/bash
# tools/sanitize_round2.sh -- see tools/sanitize_round2.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d /tmp/sanitize_r2_XXXXXX)
(cd $ROOT/tests/emu && g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-parentheses -Wno-sign-compare -o $WORK/libemu_asan.so emu.cpp)
(cd $ROOT/arriba_amd/csrc && g++ -std=c++17 -O1 -g -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-parentheses -Wno-sign-compare -o $WORK/libhost_asan.so host/*.cpp -lz)
ARRIBA_HOST_LIBRARY=$WORK/libhost_asan.so ARRIBA_EMU_LIBRARY=$WORK/libemu_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
	python $ROOT/tools/sanitize_round2.py 2>&1 | tee $WORK/log | grep -v "SAM records\|not enough chimeric reads\|early stop codon"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $WORK/log || true)"
rm -rf $WORK
