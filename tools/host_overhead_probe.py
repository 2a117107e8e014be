import time, torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import espnet_amd
from espnet_amd import ops
espnet_amd.set_precision("fp32")
x = torch.randn(10, 256, device="cuda"); W = torch.randn(256, 256, device="cuda"); b = torch.randn(256, device="cuda")
g = torch.randn(256, device="cuda")
def rate(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t) / n * 1e6, (t2 - t) / n * 1e6
with torch.no_grad():
    print("ops.linear_fwd      host %.1f us, with drain %.1f us" % rate(lambda: ops.linear_fwd(x, W, b)))
    print("ops.layernorm_fwd   host %.1f us, with drain %.1f us" % rate(lambda: ops.layernorm_fwd(x, g, b, 1e-12)))
    print("torch.add           host %.1f us, with drain %.1f us" % rate(lambda: x + 1.0))
    print("torch.empty         host %.1f us, with drain %.1f us" % rate(lambda: torch.empty(10, 256, device="cuda")))
    print("torch matmul        host %.1f us, with drain %.1f us" % rate(lambda: x @ W))
    print("stream_ptr          host %.1f us, with drain %.1f us" % rate(lambda: ops.stream_ptr()))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): ops.linear_fwd(x, W, b)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
