# CPU-only sanitizer build of the lane-code harness (make -C tests/host_emul asan).  Not shipped to the GPU box.
HIPCC ?= /opt/rocm/bin/hipcc
LOG_WG ?= 10
SAN := -fsanitize=address,undefined
emul_asan: emul.cpp $(wildcard ../../aes-gcm-128-192-256-bits_amd/csrc/*.h) aesgcm_bs.h ../../oracle/aesgcm_oracle.c
	gcc -O1 -g -std=c99 $(SAN) -c ../../oracle/aesgcm_oracle.c -o oracle_asan.o
	$(HIPCC) -O0 -g -std=c++17 --offload-arch=gfx950 --cuda-host-only -Wno-unused-value -DAESGCM_LOG_WG=$(LOG_WG) $(SAN) -fno-omit-frame-pointer -c -o emul_asan.o emul.cpp
	$(HIPCC) $(SAN) -o emul_asan emul_asan.o oracle_asan.o
