export WFT_LIB=$PWD/whisper-finetune_amd/libwft_timing.so
for i in 1 2; do
WFT_GEMM_DIAG=13 python bench.py --model base --batch 8 --hip-graph --no-extras --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tn two-buffer', d['ms_per_step'], d.get('ms_per_step_median'), d.get('final_loss'))"
WFT_GEMM_DIAG=11 python bench.py --model base --batch 8 --hip-graph --no-extras --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nt two-buffer unsplit', d['ms_per_step'], d.get('ms_per_step_median'), d.get('final_loss'))"
python bench.py --model base --batch 8 --hip-graph --no-extras --no-cpu-baseline --no-roofline --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rings', d['ms_per_step'], d.get('ms_per_step_median'), d.get('final_loss'))"
done
