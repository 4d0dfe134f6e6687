# TEST INFRASTRUCTURE ONLY: the emulator build under AddressSanitizer (CPU). GPU sanitizer builds are not allowed on the
# pool, so this file (and tools/asan_emu.sh) stays off the GPU boxes: both are listed in .gpurunignore.
#   make -C tests/emu -f asan.mk && tools/asan_emu.sh
HERE := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
SRC  := $(HERE)../../sh-assembly_amd/csrc
$(HERE)libshk_emu_asan.so: $(SRC)/shk_api.hip $(SRC)/kmer_kernels.hip $(SRC)/partition_kernels.hip $(SRC)/cqf_kernels.hip $(SRC)/merge2_kernels.hip $(SRC)/walk_kernels.hip $(SRC)/unitig_kernels.hip $(SRC)/shk_device.h $(HERE)hip/hip_runtime.h $(HERE)emu_runtime.cpp $(HERE)../../include/shk.h
	g++ -std=c++20 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -shared -Wno-unknown-pragmas -I$(HERE) -x c++ $(SRC)/shk_api.hip $(HERE)emu_runtime.cpp -o $@ -lpthread
