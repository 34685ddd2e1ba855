#!/usr/bin/env python3
"""cu_mask_probe.py -- what a CU mask on the trace stream costs, and how the mask's bits map onto the 8 XCDs.
The default scene, 1920x1080, 256 spp, scene kernel, on streams made with hipExtStreamCreateWithCUMask:
  all      : every CU                                   (the reference time)
  spread16 : 16 CUs off, chosen so that every XCD loses 2 whether bit i is CU i/8 of XCD i%8 or CU i%32 of XCD i/32
  top16    : bits 240-255 off -- 2 per XCD if the bits interleave over the XCDs, half of the last XCD if they do not
  spread32 : 32 CUs off, 4 per XCD either way
No mask leaves an XCD without CUs.  Run once on the GPU box."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa

hip = C.CDLL("libamdhip64.so.7")


def masked_stream(off_bits):
    words = [0xFFFFFFFF] * 8
    for b in off_bits:
        words[b // 32] &= ~(1 << (b % 32))
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return s


spread = lambda per_xcd: [32 * b + (b + 8 * k) % 32 for b in range(8) for k in range(per_xcd)]
masks = {"all": [], "spread16": spread(2), "top16": list(range(240, 256)), "spread32": spread(4)}
P = pa.Params(1920, 1080, 500.0, 256, 8)
for name, off in masks.items():
    st = masked_stream(off)
    with pa.Renderer(P, seed=0) as r:
        assert r.specialise()
        r.set_stream(st.value)
        r.set_timing(True)
        r.trace(256); r.synchronize()
        ts = []
        for _ in range(4):
            r.reset(); r.trace(256)
            ts.append(r.last_trace_ms()[0])
        print(f"{name:9s} {len(off):3d} CUs off: {min(ts):7.3f} ms (best of 4) = {min(ts) / (3.076 if False else 1):.3f}", flush=True)
    hip.hipStreamDestroy(st)
