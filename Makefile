# Builds everything in-tree:
#   bio_amd/libbiopileup.so   HIP kernels + C++ host code behind include/bio_pileup.h (gfx950)
#   bio_amd/bio-pileup        command-line work-alike of cmd/bio-pileup over that library
#   bio_amd/libbpsynth.so     synthetic read generator (bench/test support, host only)
#   oracle/_build/liboracle.so CPU checker (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-result
CXXFLAGS ?= -O3 -std=c++17 -fPIC -march=x86-64-v3 -Wall -pthread
CSRC := bio_amd/csrc
LIB_SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/host/*.cpp)
LIB_HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/host/*.h) include/bio_pileup.h include/bio_pileup_host.h

all: lib cli synth oracle
lib: bio_amd/libbiopileup.so
cli: bio_amd/bio-pileup
synth: bio_amd/libbpsynth.so
oracle:
	$(MAKE) -s -C oracle

bio_amd/libbiopileup.so: $(LIB_SRCS) $(LIB_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(LIB_SRCS) -lz -lpthread -ldl

bio_amd/bio-pileup: $(CSRC)/cmd/bio_pileup_main.cpp bio_amd/libbiopileup.so
	$(CXX) -O2 -std=c++17 -o $@ $(CSRC)/cmd/bio_pileup_main.cpp -Lbio_amd -lbiopileup -Wl,-rpath,'$$ORIGIN'

bio_amd/libbpsynth.so: $(CSRC)/synth.cpp include/bio_pileup.h
	$(CXX) $(CXXFLAGS) -shared -o $@ $(CSRC)/synth.cpp

clean:
	rm -f bio_amd/*.so bio_amd/bio-pileup
	$(MAKE) -C oracle clean
.PHONY: all lib cli synth oracle clean

# diagnostic build with in-kernel phase stamps (never shipped; see tools/phase_stamps.py)
stamps: bio_amd/libbiopileup_stamps.so
bio_amd/libbiopileup_stamps.so: $(LIB_SRCS) $(LIB_HDRS)
	$(HIPCC) $(HIPFLAGS) -DBP_STAMPS $(STAMPFLAGS) -shared -o $@ $(LIB_SRCS) -lz -lpthread -ldl
