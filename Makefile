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
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -ldl -o $@

oracle:
	$(MAKE) -C oracle

# C++ facade test (the reference's test_block_diagonal through include/qrkit/QRKit.hpp); needs a GPU to run
cpptest: $(LIB)
	@mkdir -p build
	g++ -O2 -std=c++14 -Wall -Iinclude tests/cpp/test_block_diagonal.cpp -Lqrkit_amd/lib -lqrkit_amd \
	    -Wl,-rpath,'$$ORIGIN/../qrkit_amd/lib' -o build/test_block_diagonal
	g++ -O2 -std=c++14 -Wall -Iinclude tests/cpp/test_compositions.cpp -Lqrkit_amd/lib -lqrkit_amd \
	    -Wl,-rpath,'$$ORIGIN/../qrkit_amd/lib' -o build/test_compositions
	g++ -O2 -std=c++14 -Wall -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/cpp/test_sharded.cpp -Lqrkit_amd/lib -lqrkit_amd \
	    -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,'$$ORIGIN/../qrkit_amd/lib' -Wl,-rpath,/opt/rocm/lib -o build/test_sharded

# CPU sanitizer targets (the GPU side has no sanitizer on this pool): the oracle as an ASan + UBSan library for
# the oracle's own tests, and the host-side integer logic of the banded solver (banded_host.hip is plain C++) with a driver on the
# reference's known answers.  tests/test_sanitizers.py runs both.
SAN := -fsanitize=address,undefined -fno-omit-frame-pointer -g
san:
	@mkdir -p build/san
	gcc -O1 -std=c99 -fPIC -ffp-contract=off -Wall -Wextra $(SAN) -shared -o build/san/libqrk_oracle_san.so oracle/qrk_oracle.c -lm
	g++ -O1 -std=c++17 -Wall $(SAN) -x c++ $(CSRC)/banded_host.hip -x c++ tests/san/banded_host_san.cpp -o build/san/banded_host_san
	ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 build/san/banded_host_san

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean cpptest san
