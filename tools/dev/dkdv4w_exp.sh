#!/bin/bash
# Developer script (this container): build libwft_eN.so variants of the dK/dV kernel with -DD4_STAMPS -DD4_EXP=N (csrc/attn.hip)
# for tools/dev/dkdv4w_stamps.py.   bash tools/dev/dkdv4w_exp.sh 0 2 10
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/whisper-finetune_amd/csrc
mkdir -p /tmp/st
make -C $C > /dev/null
rm -f $R/whisper-finetune_amd/libwft_e*.so
for e in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffinite-math-only -DD4_STAMPS -DD4_EXP=$e -c $C/attn.hip -o /tmp/st/attn_e$e.o
  (cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/whisper-finetune_amd/libwft_e$e.so misc.o norm.o gemm.o gemm_nt4w.o gemm_tn4w.o /tmp/st/attn_e$e.o audio.o optim.o f32.o)
done
ls $R/whisper-finetune_amd/*.so
