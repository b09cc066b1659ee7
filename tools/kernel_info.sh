#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / code size of a HIP object's gfx950 code object (from the code-object notes and symbol table).
#   tools/kernel_info.sh gnark-plonky2-verifier_amd/csrc/gpv_k_bn254.o [name-filter]
set -e
OBJ=$1; FILT=${2:-.}
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ $TMP/fat.bin
TGT=$($LLVM/clang-offload-bundler --list --type=o --input=$TMP/fat.bin | grep gfx950 | head -1)
$LLVM/clang-offload-bundler --type=o --targets=$TGT --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
$LLVM/llvm-readelf --notes $TMP/dev.co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.private_segment_fixed_size:/{p=$2} /\.vgpr_spill_count:/{print n, "vgpr="v, "sgpr="s, "scratch="p, "spill="$2}' | grep -E "$FILT" | sort > $TMP/a
$LLVM/llvm-readelf -s $TMP/dev.co | awk '$4=="FUNC"{print $8, "code_bytes="$3}' | sort > $TMP/b
join $TMP/a $TMP/b
rm -rf $TMP
