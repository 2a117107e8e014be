#!/usr/bin/env python3
"""What kinds of nodes do captured hipGraphs hold, and does a MEMSET node replay correctly?  (VERDICT r3 item 2 / ADVICE r3.)

Part 1 - node census: captures single torch ops and our own steps into CUDAGraph(keep_graph=True) objects and walks the nodes
with hipGraphGetNodes / hipGraphNodeGetType (espnet_amd.graphs.node_census; CUDAGraph.debug_dump writes an empty file on ROCm).
Part 2 - a minimal memset-node probe: graph = [hipMemsetAsync(buf, 0, n)] -> [copy buf to out]; buf is refilled with 7.0 OUTSIDE
the graph before every replay; prints how many bytes of `out` are non-zero after each replay (0 = the memset node did its job).
Only buffers this script owns are read or written.  Output: gpurun_out/graph_census.txt (+ the raw dot of torch.topk).
"""
import ctypes
import os
import re
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
log = open(os.path.join(OUT, "graph_census.txt"), "w")


def say(*a):
    print(*a, flush=True)
    print(*a, file=log, flush=True)


from espnet_amd import graphs  # noqa: E402


def capture(fn, keep_dot=None):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = graphs.new_graph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out, graphs.node_census(g)


def kinds(census):
    k, memsets = census
    return dict(k, memset_params=[(m["element_size"], m["width"], m["height"], m["value"]) for m in memsets]) if memsets else k


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    x320 = torch.randn(320, 5000, device=dev)
    x1 = torch.randn(1, 50000, device=dev)
    x10 = torch.randn(10, 5000, device=dev)
    idx = torch.randint(0, 10, (10,), device=dev)
    buf = torch.empty(1 << 20, device=dev)
    cases = [
        ("torch.topk [320,5000] k=10 (multi-block path)", lambda: torch.topk(x320, 10, dim=-1)),
        ("torch.topk [1,50000] k=10", lambda: torch.topk(x1, 10, dim=-1)),
        ("torch.topk [10,5000] k=15", lambda: torch.topk(x10, 15, dim=-1)),
        ("torch.sort [10,5000]", lambda: torch.sort(x10, dim=-1)),
        ("tensor.zero_() 4 MB", lambda: buf.zero_()),
        ("torch.zeros(1<<20)", lambda: torch.zeros(1 << 20, device=dev)),
        ("torch.full((10,5000), -inf)", lambda: torch.full((10, 5000), -float("inf"), device=dev)),
        ("cumsum [249]", lambda: torch.cumsum(x10[0, :249], 0)),
        ("logsumexp [10,2]", lambda: torch.logsumexp(x10[:, :2], dim=-1)),
        ("scatter_ [10,5000]", lambda: torch.full((10, 5000), -1e10, device=dev).scatter_(1, idx.view(10, 1), x10[:, :1])),
        ("index_select [10,5000]", lambda: x10.index_select(0, idx)),
        ("tensor.copy_ (d2d 4 MB)", lambda: buf.clone()),
    ]
    say("== part 1: node kinds per captured op")
    for name, fn in cases:
        keep = os.path.join(OUT, "graph_topk320.dot") if name.startswith("torch.topk [320") else None
        _, _, text = capture(fn, keep)
        say("%-48s %s" % (name, kinds(text)))

    say("== part 2: memset node probe (hipMemsetAsync captured, buffer refilled with 7.0 before every replay)")
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetD32Async.restype = ctypes.c_int
    for nbytes in (16, 1024, 4096, 1 << 20):
        for form in ("memset8", "memset32", "fill_kern"):
            b = torch.empty(nbytes // 4, device=dev)
            o = torch.empty_like(b)

            def work():
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                if form == "fill_kern":            # control: the same block zeroed by a kernel node
                    b.fill_(0.0)                   # torch's fill kernel
                    rc = 0
                elif form == "memset8":
                    rc = hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr()), 0, nbytes, st)
                else:
                    rc = hip.hipMemsetD32Async(ctypes.c_void_p(b.data_ptr()), 0, nbytes // 4, st)
                assert rc == 0, rc
                o.copy_(b)
                return o
            g, _, text = capture(work)
            res = []
            for r in range(4):
                b.fill_(7.0)
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                res.append(int((o.view(torch.int32) != 0).sum()) * 4)
            words = " ".join("%08x" % (v & 0xffffffff) for v in o.view(torch.int32)[:8].tolist())
            say("%-9s %8d bytes: nodes %s ; non-zero bytes of the copy after replay 0..3: %s ; first words after replay 3: %s"
                % (form, nbytes, kinds(text), res, words))

    if len(sys.argv) > 1 and sys.argv[1] == "steps":
        say("== part 3: our captured steps")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_model import c2width_setup
        from espnet_amd.nets.beam_search import BeamSearch
        SW, model, gold, encs = c2width_setup()
        bs = BeamSearch(model.scorers(), dict(decoder=0.7, ctc=0.3), 10, 5000, model.sos, model.eos, pre_beam_score_key="full")
        with torch.no_grad():
            Ts = [int(e.shape[0]) for e in encs]
            C_ = bs._batch_consts(encs, Ts, [20] * 3, 256, 22, always_mask=True)
            S = bs._batch_state0(C_)
            for i in range(2):
                S, _ = bs._batch_step(i, C_, S)                     # lazily built tensors outside the capture
            torch.cuda.synchronize()
            S2 = dict(S)
            _, _, text = capture(lambda: bs._batch_step(2, C_, S2), os.path.join(OUT, "graph_beam_step.dot"))
        say("%-48s %s" % ("beam step (3 utterances x beam 10, V 5000)", kinds(text)))
    say("done")


if __name__ == "__main__":
    main()
