# Builds the in-tree C-ABI HIP library for gfx950 (cross-compiles without a GPU).
HIPCC ?= hipcc
ARCH  ?= gfx950
SRC   := $(wildcard tvqaplus_amd/csrc/*.hip)
OBJ   := $(patsubst tvqaplus_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := tvqaplus_amd/libstage_hip.so
# second build of the streaming GEMMs with TWO bf16 terms per fp32 operand (hi + mid: three products instead of six, every
# product accurate to ~2^-17 instead of fp32's 2^-24): the opt-in fast mode, selected at run time by STAGE_GEMM_TERMS=2
# (tvqaplus_amd/_lib.py) -- the default library keeps the exact 3-term split
LIB2  := tvqaplus_amd/libstage_hip_t2.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value

all: $(LIB) $(LIB2)

build/%.o: tvqaplus_amd/csrc/%.hip tvqaplus_amd/csrc/common.h include/stage_hip.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

build/t2_gemm_stream.o: tvqaplus_amd/csrc/gemm_stream.hip tvqaplus_amd/csrc/common.h include/stage_hip.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -DSTAGE_GEMM_TERMS=2 -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

$(LIB2): $(OBJ) build/t2_gemm_stream.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter-out build/gemm_stream.o,$(OBJ)) build/t2_gemm_stream.o

clean:
	rm -rf build $(LIB) $(LIB2)
.PHONY: all clean
