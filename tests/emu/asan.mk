# CPU-only sanitizer build of the emulated kernels (never shipped to the GPU box: see .gpurunignore)
include $(dir $(abspath $(lastword $(MAKEFILE_LIST))))Makefile
# same thing under AddressSanitizer (static LDS arrays and heap "device" buffers are checked)
$(HERE)libshk_emu_asan.so: $(SRC)/shk_api.hip $(SRC)/kmer_kernels.hip $(SRC)/partition_kernels.hip $(SRC)/cqf_kernels.hip $(SRC)/shk_device.h $(HERE)hip/hip_runtime.h $(HERE)emu_runtime.cpp $(HERE)../../include/shk.h
	g++ -std=c++20 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -shared -w -I$(HERE) -x c++ $(SRC)/shk_api.hip $(HERE)emu_runtime.cpp -o $@ -lpthread
asan: $(HERE)libshk_emu_asan.so
