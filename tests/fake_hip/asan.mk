# The host side of the library + the fake runtime + the threaded driver (mt_drive.cpp) under the address and undefined-behaviour sanitizers: one executable, the
# sanitizer runtime linked in (no preload needed).  CPU only: make -C tests/fake_hip -f asan.mk.  Not shipped to the GPU box (the pool refuses sanitizer builds).
HIPCC ?= /opt/rocm/bin/hipcc
CSRC  := ../../aes-gcm-128-192-256-bits_amd/csrc
SAN   ?= -fsanitize=address,undefined
TAG   ?= asan
FLAGS := -DAESGCM_LOG_WG=10 -O1 -g -fno-omit-frame-pointer -std=c++17 --offload-arch=gfx950 --cuda-host-only -Wno-unused-value -Wno-unused-function $(SAN)
OBJS  := host.$(TAG).o abi.$(TAG).o comm.$(TAG).o fakehip.$(TAG).o mt_drive.$(TAG).o
mt_drive_$(TAG): $(OBJS)
	$(HIPCC) $(SAN) -rdynamic -o $@ $(OBJS) -ldl -lpthread
host.$(TAG).o: $(CSRC)/aesgcm_host.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) $(FLAGS) -c -o $@ $<
abi.$(TAG).o: $(CSRC)/aesgcm_abi.hip $(wildcard $(CSRC)/*.h)
	$(HIPCC) $(FLAGS) -c -o $@ $<
# the fake RCCL lives in the executable itself: dlopen(NULL) and -rdynamic
comm.$(TAG).o: $(CSRC)/aesgcm_comm.hip
	$(HIPCC) $(FLAGS) '-DAESGCM_TEST_RCCL_LIB=(const char *)0' -c -o $@ $<
fakehip.$(TAG).o: fakehip.cpp $(wildcard $(CSRC)/*.h)
	$(HIPCC) $(FLAGS) -x hip -c -o $@ $<
mt_drive.$(TAG).o: mt_drive.cpp ../../include/aesgcm.h
	$(HIPCC) $(FLAGS) -x hip -c -o $@ $<
clean:
	rm -f *.asan.o *.tsan.o mt_drive_asan mt_drive_tsan
