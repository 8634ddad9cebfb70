# Plain-make build of the C-ABI library and the C host program, for users who do not want
# Python in the loop (python -m adsbdec_amd._build runs the same commands; keep them in step).
# hipcc cross-compiles the gfx950 code object without a GPU.
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
# -ffp-contract=off: the reference arithmetic is binary32 multiply THEN add
# -amdgpu-atomic-optimizer-strategy=None: see adsbdec_amd/_build.py (the LDS survivor-queue atomics stay per lane)
HIPFLAGS ?= --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -Xarch_host -mavx2
CSRC     := adsbdec_amd/csrc
LIBDIR   := adsbdec_amd/lib
LIB      := $(LIBDIR)/libadsbdec_amd.so
CLI      := $(LIBDIR)/adsbdec_amd_cli
# dependencies: every compile writes <object>.d (-MMD), included at the bottom -- no hand-kept header list to forget a file in

all: $(LIB) $(CLI)

$(LIBDIR)/%.hip.o: $(CSRC)/%.hip
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -MMD -MF $@.d -c $< -o $@

$(LIBDIR)/format.c.o: $(CSRC)/format.c
	@mkdir -p $(LIBDIR)
	$(CC) -O2 -fPIC -Wall -MMD -MF $@.d -c $< -o $@

$(LIBDIR)/%.cpp.o: $(CSRC)/%.cpp
	@mkdir -p $(LIBDIR)
	g++ -O2 -fPIC -std=c++17 -Wall -Wextra -pthread -MMD -MF $@.d -c $< -o $@

$(LIB): $(LIBDIR)/scan_kernel.hip.o $(LIBDIR)/decoder.hip.o $(LIBDIR)/format.c.o $(LIBDIR)/multi.cpp.o $(LIBDIR)/host_abi.cpp.o $(LIBDIR)/numa.cpp.o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -lm -lpthread

$(CLI): $(CSRC)/cli/adsbdec_amd_cli.c $(CSRC)/cli/sink.c $(CSRC)/cli/sink.h $(LIB) include/adsbdec_amd.h
	$(CC) -O2 -Wall -o $@ $(CSRC)/cli/adsbdec_amd_cli.c $(CSRC)/cli/sink.c -Iinclude -L$(LIBDIR) -ladsbdec_amd -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

# test infrastructure (never linked into the library): the CPU oracle, and the reference
# objects it is pinned against when /root/reference is present
oracle:
	$(MAKE) -C oracle

check: all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -f $(LIBDIR)/*.o $(LIBDIR)/*.o.d $(LIB) $(CLI)

-include $(wildcard $(LIBDIR)/*.o.d)

.PHONY: all oracle check clean
