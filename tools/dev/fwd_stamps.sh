#!/bin/bash
# Developer script (this container): build whisper-finetune_amd/libwft_fwdstamps.so = the shipped objects with attn.hip compiled
# -DFWD_STAMPS (csrc/attn.hip: s_memtime sums per phase of a key tile in attn_fwd_kernel) for tools/dev/fwd_stamps.py.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/whisper-finetune_amd/csrc
mkdir -p /tmp/st
make -C $C > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffinite-math-only -DFWD_STAMPS ${FWD_EXTRA} -c $C/attn.hip -o /tmp/st/attn_fwdstamps.o
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/whisper-finetune_amd/libwft_fwdstamps${FWD_TAG}.so misc.o norm.o gemm.o gemm_nt4w.o gemm_tn4w.o /tmp/st/attn_fwdstamps.o audio.o optim.o f32.o)
ls -la $R/whisper-finetune_amd/libwft_fwdstamps${FWD_TAG}.so
