#!/usr/bin/env python3
"""Does the front-end launch run slower when the channel decoder runs between its launches (as in the bench's step)?
Same buffers (dabgpu_alloc_frame_buffers), front end timed by the library's own events around the kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import torch, dabgpu
dev = torch.device("cuda", 0)
E, F = 64, 256; n = E * F; L, NB = dabgpu.NB_FRAME_SAMPLES, dabgpu.NB_FRAME_BITS
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
c = dabgpu.Context(0, n)
d_iq, d_soft, table, kept = c.alloc_frame_buffers(n, L, 4)
print("probe", [[round(float(x), 3) for x in r] for r in table], kept)
fo = torch.zeros((n,), dtype=torch.float32, device=dev); cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); crc = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3); msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
def ofdm(): c.ofdm_demod_frames_dev(d_iq + 2656 * 8, L, n, fo.data_ptr(), d_soft, cyc.data_ptr(), None, s)
def dec(): c.decode_frames_dev(d_soft, NB, E, F, fib.data_ptr(), crc.data_ptr(), [sc], None, None, [msc.data_ptr()], s)
def measure(label, seq, rounds=10):
    for f in seq * 2: f()
    torch.cuda.synchronize(); c.set_timing(True)
    for _ in range(rounds):
        for f in seq: f()
    torch.cuda.synchronize()
    ms, k = c.mean_kernel_ms(0)
    c.set_timing(False)
    print("%-46s front end %.3f ms (mean of %d launches)" % (label, ms, k))
for rep in range(2):
    measure("front end back to back", [ofdm])
    measure("front end, decoder, front end, decoder ...", [ofdm, dec])
    measure("front end x2, decoder x2", [ofdm, ofdm, dec, dec])
    measure("front end, 2 ms idle (sleep kernel)", [ofdm, lambda: torch.cuda._sleep(4000000)])
