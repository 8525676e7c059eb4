"""Developer script (GPU box, WFT_GEMM_DIAG=12): where does an NT256 tile's time go?  s_memrealtime stamps (100 MHz) of waves 0 / 4
of every persistent workgroup: kernel entry, per tile {main loop start, main loop end, after the end barrier, epilogue end}."""
import ctypes as C, os, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__))); import _timing_lib  # noqa: E702 (WFT_LIB -> libwft_timing.so)
DIAG = os.environ.get("WFT_GEMM_DIAG", "12")
os.environ["WFT_GEMM_DIAG"] = DIAG
import torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "whisper-finetune_amd"))
from whisper_finetune.engine import kernels as K, lib as L
import numpy as np
dev = torch.device("cuda:0")

def run(M, N, Kd, res=False):
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    r = torch.randn(M, N, device=dev).bfloat16() if res else None
    st = torch.zeros(256 * 2 * 160, dtype=torch.int64, device=dev)
    args = L.GemmArgs()
    args.A, args.lda = a.data_ptr(), Kd
    args.B, args.ldb = b.data_ptr(), Kd
    args.C, args.ldc = o.data_ptr(), N
    if res: args.residual, args.ldr = r.data_ptr(), N
    args.alpha, args.beta = 1.0, 1.0
    args.M, args.N, args.K, args.batch = M, N, Kd, 1
    args.workspace, args.workspace_bytes = st.data_ptr(), st.numel() * 8
    for _ in range(3):
        L.check(L.load().wft_gemm_nt_bf16(C.byref(args), L.stream_ptr()), "nt")
    torch.cuda.synchronize()
    st.zero_()
    L.check(L.load().wft_gemm_nt_bf16(C.byref(args), L.stream_ptr()), "nt")
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 2, 160).astype(np.float64) / 100.0  # us
    t0 = s[:, :, 0].min()
    nin = Kd // 32 // 8 - 1
    per = 4 + nin
    ntile = int(((s[0, 0] > 0).sum() - 1) // per)
    print(f"--- M={M} N={N} K={Kd} res={res}: {ntile} tiles per workgroup; kernel entry spread {s[:, 0, 0].max() - t0:.1f} us")
    for g, name in ((0, "group A"), (1, "group B")):
        xx = s[:, g, 1:1 + per * ntile].reshape(256, ntile, per) - t0
        x = np.concatenate([xx[:, :, :1], xx[:, :, -3:]], axis=2)
        seg = np.diff(xx[:, :, :nin + 2], axis=2)  # 8-slab segments of the main loop
        main = x[:, :, 1] - x[:, :, 0]; wait = x[:, :, 2] - x[:, :, 1]; epi = x[:, :, 3] - x[:, :, 2]
        gap = x[:, 1:, 0] - x[:, :-1, 3]
        print(f"{name}: first main-loop start {x[:, 0, 0].mean():.1f} us after entry; last epilogue end mean {x[:, -1, 3].mean():.1f} max {x[:, -1, 3].max():.1f}")
        for k in range(ntile):
            print(f"  tile {k}: main {main[:, k].mean():6.2f} (min {main[:, k].min():6.2f} max {main[:, k].max():6.2f})  end-barrier {wait[:, k].mean():5.2f}  epilogue {epi[:, k].mean():5.2f} (min {epi[:, k].min():5.2f} max {epi[:, k].max():5.2f})"
                  + (f"  gap to next {gap[:, k].mean():5.2f}" if k + 1 < ntile else ""))

        if g == 0:
            mm = main.mean(1)  # per workgroup
            print("   per XCD group (blockIdx % 8) mean main:", " ".join(f"{mm[x::8].mean():.2f}" for x in range(8)))
            print("   per local index (blockIdx // 8) mean main:", " ".join(f"{mm[8 * j:8 * j + 8].mean():.1f}" for j in range(32)))
            # tile coordinates (band order of gemm.hip) of each workgroup's tiles: does the slow set follow the column / row?
            tiles_r, tiles_c = (M + 255) // 256, N // 256
            ntl = tiles_r * tiles_c
            def coords(t):
                q, r8, x = ntl >> 3, ntl & 7, t & 7
                sid = (x * (q + 1) if x < r8 else r8 * (q + 1) + (x - r8) * q) + (t >> 3)
                W = 5; band = sid // (tiles_r * W); c0 = band * W; w = min(W, tiles_c - c0); r = sid - band * tiles_r * W
                return r // w, c0 + r % w, sid
            by_c = {}; by_pos = {}
            for wg in range(256):
                for k in range(ntile):
                    t = wg + 256 * k
                    if t >= ntl: continue
                    tr, tc, sid = coords(t)
                    by_c.setdefault(tc, []).append(main[wg, k]); by_pos.setdefault((wg // 8) % 32, []).append(main[wg, k])
            print("   8-slab segments, all tiles: mean", " ".join(f"{v:.2f}" for v in seg.mean((0, 1))), "| slowest 10% of tiles:", " ".join(f"{v:.2f}" for v in seg.reshape(-1, nin + 1)[np.argsort(main.reshape(-1))[-main.size // 10:]].mean(0)),
                  "| fastest 10%:", " ".join(f"{v:.2f}" for v in seg.reshape(-1, nin + 1)[np.argsort(main.reshape(-1))[:main.size // 10]].mean(0)))
            print("   mean main by tile column:", " ".join(f"{c}:{np.mean(v):.1f}" for c, v in sorted(by_c.items())))
            slow = [(wg, k) for wg in range(0, 256, 8) for k in range(ntile) if wg + 256 * k < ntl and main[wg, k] > np.median(main) * 1.08]
            print("   slow tiles of XCD group 0 (local index, round -> row, col):", " ".join(f"({wg // 8},{k}->{coords(wg + 256 * k)[0]},{coords(wg + 256 * k)[1]})" for wg, k in slow[:60]))
            print("   tile-by-tile main of workgroups 0, 8, 16, 24, 1, 9:", " | ".join(" ".join(f"{v:.1f}" for v in main[w]) for w in (0, 8, 16, 24, 1, 9)))

import itertools
for shp in ((65536, 1280, 1280, False), (65536, 1280, 1280, True)):
    run(*shp)
