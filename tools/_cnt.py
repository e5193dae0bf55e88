import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
dev = torch.device("cuda", 0)
n = 256
for name, gen in (("noise", lambda: torch.randint(-127, 128, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)),
                  ("small noise", lambda: torch.randint(-8, 9, (n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev)),
                  ("zeros", lambda: torch.zeros((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev))):
    soft = gen()
    fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
    c = dabgpu.Context(0, 8)
    c.fic_decode_dev(soft.data_ptr(), dabgpu.NB_FRAME_BITS, n, fib.data_ptr(), ok.data_ptr(), None); c.sync()
    f = fib.cpu().numpy().reshape(n * 4, 96)
    print("FIC", name, "passes hist", np.bincount(f[:, 0], minlength=8)[:12], "max", f[:, 0].max(), " mismatching pairs after warm-up: mean %.2f max %d" % (f[:, 1].mean(), f[:, 1].max()))
    sc = dabgpu.subchannel(0, 64, level=3)
    msc = torch.zeros((1, n * 4, 192), dtype=torch.uint8, device=dev)
    hin = torch.randint(-127, 128, (1, 15, sc.length * 64), dtype=torch.int8, device=dev) if name != "zeros" else torch.zeros((1, 15, sc.length * 64), dtype=torch.int8, device=dev)
    hout = torch.zeros_like(hin)
    c.msc_decode_dev(sc, soft.data_ptr(), dabgpu.NB_FRAME_BITS, 1, n, hin.data_ptr(), hout.data_ptr(), msc.data_ptr(), None); c.sync()
    m = msc.cpu().numpy().reshape(n * 4, 192)
    print("MSC", name, "passes hist", np.bincount(m[:, 0], minlength=8)[:12], "max", m[:, 0].max(), " mismatching pairs after warm-up: mean %.2f max %d" % (m[:, 1].mean(), m[:, 1].max()))
    c.close()
