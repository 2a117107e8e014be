#!/usr/bin/env python3
"""s_memtime stamps of one up / one down wave of the fused FFN kernels (diagnostic build tools/libffn_stamp.so =
csrc/ffn_f32.hip compiled with -DFFN_STAMP): where a step's cycles go.  Stamp k of step s:
0 step start, 1 first-half MFMAs issued, 2 epilogue quarter done, 3 tile stores issued, 4 in front of the barrier,
5 behind it, 6 second-half MFMAs issued (up waves), 7 second epilogue quarter done."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from espnet_amd import _lib, ops  # noqa: E402

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libffn_stamp.so"))
M, D, F = 7968, 256, 2048
dev = "cuda"
x = torch.randn(M, D, device=dev); w1 = torch.randn(F, D, device=dev) * 0.05; b1 = torch.zeros(F, device=dev)
w2 = torch.randn(D, F, device=dev) * 0.05; b2 = torch.zeros(D, device=dev); R = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev)
ops.manual_seed(1)
out = torch.empty(M, D, device=dev); f = torch.empty(M, F, device=dev); h = torch.empty(M, F, device=dev)
dz = torch.empty(M, F, device=dev); dx = torch.empty(M, D, device=dev)
pf = ops._ffn_desc(x, w1, b1, w2, b2, R, out, f, h, ops.ACT_SWISH, 0.5, (0.1, 11, 0.1, 12))
pb = ops._ffn_desc(dy, w1, None, w2, None, None, dx, f, dz, ops.ACT_NONE, 0.5, (0.0, 0, 0.0, 0))
for _ in range(5):
    assert L.eamd_ffn_fwd(C.byref(pf), _lib.stream_ptr()) == 0
    assert L.eamd_ffn_bwd(C.byref(pb), _lib.stream_ptr()) == 0
torch.cuda.synchronize()
buf = (C.c_uint64 * (2 * 2 * 8 * 8))()
assert L.eamd_ffn_debug_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 2, 8, 8).astype(np.int64)
for kern, kn in ((0, "fwd"), (1, "bwd")):
    for role, rn in ((0, "up  "), (1, "down")):
        t0 = a[kern, role, 0, 0]
        print("%s %s wave: stamps relative to the body's first (100 MHz ticks x ?) per step" % (kn, rn))
        for s in range(8):
            r = a[kern, role, s] - t0
            print("   s=%d  " % s + "  ".join("%6d" % v for v in r))
        print("   body length %d" % (a[kern, role, 7].max() - t0))
