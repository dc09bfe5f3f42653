#!/bin/bash
# disassemble the gfx950 code object of one translation unit: tools/disasm.sh conv_cl16 [outdir] -> <outdir>/<unit>.s
U=${1:-conv_cl16}; O=${2:-/tmp/dis}; mkdir -p $O
cp "$(dirname "$0")/../dcvgan_amd/csrc/obj/$U.o" $O/$U.o
( cd $O && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $U.o > /dev/null 2>&1 )
f=$(ls $O/$U.o.*gfx950* 2>/dev/null | head -1)
[ -n "$f" ] || { echo "no gfx950 bundle in $U.o"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 "$f" > $O/$U.s
echo $O/$U.s
