#!/bin/bash
# Developer script (this container): libwft_fwdabl<N>.so for N in "$@" = the shipped objects with attn.hip compiled -DFWD_ABL=N
# (csrc/attn.hip: attn_fwd_kernel with one ingredient removed; timing only).   bash tools/dev/fwd_abl.sh 1 2 3 4 5 6 7 8
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/whisper-finetune_amd/csrc
mkdir -p /tmp/st
make -C $C > /dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffinite-math-only -DFWD_ABL=$n -c $C/attn.hip -o /tmp/st/attn_abl$n.o
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/whisper-finetune_amd/libwft_fwdabl$n.so misc.o norm.o gemm.o gemm_nt4w.o gemm_tn4w.o /tmp/st/attn_abl$n.o audio.o optim.o f32.o)
done
ls $R/whisper-finetune_amd/libwft_fwdabl*.so
