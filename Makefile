# Builds the gfx950 shared library behind the C ABI (include/qrkit_amd.h) and the CPU oracle.
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
HIPFLAGS  ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC      := qrkit_amd/csrc
OBJDIR    := build/obj
LIB       := qrkit_amd/lib/libqrkit_amd.so
SRCS      := $(wildcard $(CSRC)/*.hip)
OBJS      := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))

all: $(LIB) oracle cpptest

$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/qrkit_amd.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the exact-arithmetic path must round like a scalar evaluation of Eigen's algorithm: never contract a*b+c into an FMA
$(OBJDIR)/bdqr_exact.o: HIPFLAGS += -ffp-contract=off

$(LIB): $(OBJS)
	@mkdir -p qrkit_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

oracle:
	$(MAKE) -C oracle

# C++ facade test (the reference's test_block_diagonal through include/qrkit/QRKit.hpp); needs a GPU to run
cpptest: $(LIB)
	@mkdir -p build
	g++ -O2 -std=c++14 -Wall -Iinclude tests/cpp/test_block_diagonal.cpp -Lqrkit_amd/lib -lqrkit_amd \
	    -Wl,-rpath,'$$ORIGIN/../qrkit_amd/lib' -o build/test_block_diagonal
	g++ -O2 -std=c++14 -Wall -Iinclude tests/cpp/test_compositions.cpp -Lqrkit_amd/lib -lqrkit_amd \
	    -Wl,-rpath,'$$ORIGIN/../qrkit_amd/lib' -o build/test_compositions

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean cpptest
