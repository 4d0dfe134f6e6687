#!/bin/bash
# The emulator tests and a randomised run with the kernels compiled under AddressSanitizer (CPU build, tests/emu).
# The sanitized library takes the place of tests/emu/libshk_emu.so for the duration of the run.
set -e
cd "$(dirname "$0")/.."
make -s -C tests/emu && make -s -C tests/emu -f asan.mk
cp tests/emu/libshk_emu.so /tmp/libshk_emu_plain.so
trap 'cp /tmp/libshk_emu_plain.so tests/emu/libshk_emu.so' EXIT
cp tests/emu/libshk_emu_asan.so tests/emu/libshk_emu.so && touch tests/emu/libshk_emu.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
python -m pytest tests/test_emu_kernels.py -x -q -k "not randomised"
python tools/fuzz_gpu.py --emu --cases ${1:-40} --seed 9 --max-qb 13
