#!/usr/bin/env python3
"""GPU box: front end of batch k and channel decoder of batch k-1 on two streams, with launches SIZED to share a CU
(DABGPU_FLAG_CORESIDENT: the front end asks for 56 KB of LDS per workgroup, so two fit a CU instead of three, and the
grouped lane decoder's forward pass for exactly its own 41 KB, so one of its workgroups fits beside them; register file:
2 x 168 + 136 VGPRs per SIMD lane of 512).  Round 2's two-stream overlap lost because the decoder's workgroups took the
LDS and registers the front end's third workgroup needed, unevenly over the CUs.
Needs a library built with tools/patches/coresident_flag.patch applied (`git apply tools/patches/coresident_flag.patch`,
`make -C sdrplusplus-dab-radio-plugin_amd/csrc`): the flag is NOT in the product -- measured (profiles/r03_overlap_coresident.txt):
the front end alone loses 6 % at two workgroups per CU, and the two kernels together take as long as one after the
other (7.46 ms vs 7.13 ms serial): both are compute-heavy enough that sharing the SIMDs gains nothing.
usage: tools/overlap_coresident.py [ensembles] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sdrplusplus-dab-radio-plugin_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, dabgpu
E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = E * F
dev = torch.device("cuda", 0)
L = dabgpu.NB_FRAME_SAMPLES
iq = torch.empty((E, F * L, 2), dtype=torch.float32, device=dev).normal_()
soft = [torch.empty((n, dabgpu.NB_FRAME_BITS), dtype=torch.int8, device=dev) for _ in range(2)]
cyc = torch.zeros((n, 76, 2), dtype=torch.float32, device=dev)
fib = torch.zeros((n, 12, 32), dtype=torch.uint8, device=dev); ok = torch.zeros((n, 12), dtype=torch.uint8, device=dev)
sc = dabgpu.subchannel(0, 64, level=3)
msc = torch.zeros((E, F * 4, 192), dtype=torch.uint8, device=dev)
hist = [torch.zeros((E, 15, sc.length * 64), dtype=torch.int8, device=dev) for _ in range(2)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def run(flags_fe, flags_dec, mode, reps=10):
    octx = dabgpu.Context(0, n, flags=flags_fe); octx.streams_reset(E)
    dctx = dabgpu.Context(0, n, flags=flags_dec)

    def demod(k, st):
        octx.ofdm_demod_streams_dev(iq.data_ptr() + 2656 * 8, L, E, F, 0.5, soft[k & 1].data_ptr(), cyc.data_ptr(), None, st.cuda_stream)

    def decode(k, st):
        dctx.decode_frames_dev(soft[k & 1].data_ptr(), dabgpu.NB_FRAME_BITS, E, F, fib.data_ptr(), ok.data_ptr(), [sc],
                               [hist[k & 1].data_ptr()], [hist[(k & 1) ^ 1].data_ptr()], [msc.data_ptr()], st.cuda_stream)
    evA = [torch.cuda.Event() for _ in range(2)]; evB = [torch.cuda.Event() for _ in range(2)]

    def step(k):
        if mode == "serial":
            demod(k, sA); decode(k, sA)
        elif mode == "fe":
            demod(k, sA)
        elif mode == "dec":
            decode(k, sA)
        else:                                                # pipelined: FE(k) on A beside decode(k-1)... here decode(k) follows FE(k) on B
            if k >= 2:
                sA.wait_event(evB[k & 1])                    # soft[k & 1] was decoded two steps ago
            demod(k, sA)
            evA[k & 1].record(sA)
            sB.wait_event(evA[k & 1])
            decode(k, sB)
            evB[k & 1].record(sB)
    for k in range(4): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps): step(4 + k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    octx.close(); dctx.close()
    return dt


C = getattr(dabgpu, 'FLAG_CORESIDENT', 1 << 3)
print("step = front end + FIC/MSC decode of %d frames; ms per step" % n)
for rep in range(2):
    print("serial, default launches:                          %.3f" % run(0, 0, "serial"))
    print("front end alone: default %.3f | two workgroups per CU %.3f" % (run(0, 0, "fe"), run(C, 0, "fe")))
    print("decoder alone:   default %.3f | exact LDS %.3f" % (run(0, 0, "dec"), run(0, C, "dec")))
    print("two streams, default launches (round 2's experiment): %.3f" % run(0, 0, "pipe"))
    print("two streams, co-resident launches:                 %.3f" % run(C, C, "pipe"))
    print("two streams, front end co-resident only:           %.3f" % run(C, 0, "pipe"))
    print("two streams, decoder exact LDS only:               %.3f" % run(0, C, "pipe"))
