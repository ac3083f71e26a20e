# Builds the in-tree C-ABI HIP library for gfx950 (cross-compiles without a GPU).
HIPCC ?= hipcc
ARCH  ?= gfx950
SRC   := $(wildcard tvqaplus_amd/csrc/*.hip)
OBJ   := $(patsubst tvqaplus_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := tvqaplus_amd/libstage_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value

all: $(LIB)

build/%.o: tvqaplus_amd/csrc/%.hip tvqaplus_amd/csrc/common.h include/stage_hip.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)
.PHONY: all clean
