P=whisper-finetune_amd
for i in 1 2 3; do
for l in libwft.so libwft_fe1.so libwft_fe2.so libwft_fe3.so; do WFT_TIME_PRE=1 WFT_LIB=$PWD/$P/$l python tools/dev/attn_fwd_time.py; done
done 2>&1 | grep -v amdgpu.ids
