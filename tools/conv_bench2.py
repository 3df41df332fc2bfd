"""Per-layer timing of the DCGAN k4 s2 p1 layers (F / Dg / Wg) with the shader clock warmed up: the clock ramps from
~1.9 to ~2.39 GHz over the first ~40 ms of load (tools/igemm2_probe.hip, in-kernel s_memtime / s_memrealtime), so
every measurement is preceded by 0.3 s of the same launches and nothing synchronises in between.

    python tools/conv_bench2.py [bs] [ops: f,d,w] [--stats]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_gan_zoo_amd import functional as F      # noqa: E402

g = F.K4S2P1
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ops = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f", "d", "w"]
stats = "--stats" in sys.argv


def timeit(fn, n=30, warm_s=0.3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            fn()
        torch.cuda.current_stream().synchronize() if False else None
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


layers = [('D.b1', 64, 32, 128), ('D.b2', 128, 16, 256), ('D.b3', 256, 8, 512), ('G.b2', 512, 8, 1024),
          ('G.b3', 256, 16, 512), ('G.b4', 128, 32, 256)]
print('bs', bs, 'GZ_NO_IGEMM2', os.environ.get('GZ_NO_IGEMM2'), 'stats', stats)
for name, C, H, K in layers:
    x = torch.randn(bs, C, H, H, device='cuda')
    w = torch.randn(K, C, 4, 4, device='cuda') * 0.05
    gy = torch.randn(bs, K, H // 2, H // 2, device='cuda')
    fl = 2.0 * bs * (H // 2) ** 2 * K * C * 16
    out = '%-6s C%4d H%3d K%5d GF %6.1f |' % (name, C, H, K, fl / 1e9)
    for op in ops:
        if op == "f":
            fn = (lambda: F._conv_fwd_stats_raw(x, w, g)) if stats else (lambda: F._conv_fwd_raw(x, w, None, g, 0, 0.))
            lab = F.tile_label(0, (bs, C, H, H, K, H // 2, H // 2), g) if hasattr(F, "tile_label") else ""
        elif op == "d":
            fn = (lambda: F._conv_dgrad_stats_raw(gy, w, g, (H, H))) if stats else (lambda: F._conv_dgrad_raw(gy, w, None, g, (H, H), 0, 0.))
            lab = ""
        else:
            fn = lambda: F._conv_wgrad_raw(x, gy, g)      # noqa: E731
            lab = ""
        t = timeit(fn)
        out += ' %s %7.3f ms %6.1f TF |' % (op, t, fl / t / 1e9)
        if "--stamps" in sys.argv:
            import ctypes
            import numpy as np
            from lightning_gan_zoo_amd._lib import lib
            dll = lib.load()
            if hasattr(dll, "gz_debug_read_stamps"):
                nwg = 2048
                buf = np.zeros((nwg, 8), dtype=np.uint64)
                dll.gz_debug_read_stamps(ctypes.c_void_p(buf.ctypes.data), nwg)
                b = buf[buf[:, 3] > 0].astype(np.float64)
                if len(b):
                    ideal = b[:, 6] * 8 * b[:, 7] * 64
                    clk = (b[:, 3] - b[:, 0]) / (b[:, 5] - b[:, 4]) * 100
                    out += (' [wg %d: prologue %.0f, loop %.0f (1-wave ideal %.0f), epilogue %.0f cyc, clock %.0f MHz, '
                            'span %.0f us]' % (len(b), (b[:, 1] - b[:, 0]).mean(), (b[:, 2] - b[:, 1]).mean(), ideal.mean(),
                                               (b[:, 3] - b[:, 2]).mean(), clk.mean(), (b[:, 5].max() - b[:, 4].min()) / 100))
    print(out, flush=True)
