#!/bin/bash
# Device ISA of one kernel translation unit as text, one instruction per line without addresses or encodings -- to check that an edit
# which should not change the generated code (removing compile-time-dead branches, moving code) did not:
#   tools/isa_dump.sh nerf-ca_amd/csrc/nca_kernels_bf16.hip /tmp/after.s ["extra flags"]   &&   diff /tmp/before.s /tmp/after.s
set -e
SRC=$1; OUT=$2; EXTRA=$3
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $EXTRA --cuda-device-only -c "$SRC" -o $T/x.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/x.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/x.elf
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/x.elf | sed -E 's#//.*$##; s/[[:space:]]+$//' | grep -v "^$" > "$OUT"
rm -rf $T
