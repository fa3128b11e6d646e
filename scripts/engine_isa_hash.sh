#!/bin/bash
# Hash of the gfx950 ISA of the SHIPPED decode-engine object (mistral-inference_amd/build/decode_engine.o; or $1).  The default
# compile of csrc/decode_engine.hip must stay byte-identical when the file gains `#if ENG_WIDE` code: the kernel's speed moves
# by several per cent with any change of its instruction stream (profiles/EXPERIMENTS.md).  Round 3/4 value:
#   6a25db5be9eed9576d269c0111c1b1f2
set -e
O=${1:-$(dirname "$0")/../mistral-inference_amd/build/decode_engine.o}
T=$(mktemp -d)
cp "$O" "$T/x.o"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$T"/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | grep -v "file format" | md5sum | cut -d' ' -f1
rm -rf "$T"
