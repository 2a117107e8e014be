import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import espnet_amd
from espnet_amd import ops, rnn_functional as R
DEV = torch.device("cuda")
B, H, T, ndir = 32, 1024, 40, 2
g = torch.Generator().manual_seed(1)
ws = [(torch.randn(4 * H, H, generator=g).mul(1.0 / H ** 0.5).to(DEV), torch.randn(4 * H, generator=g).mul(0.1).to(DEV)) for _ in range(ndir)]
gxs = [torch.randn(T, B, 4 * H, generator=g).to(DEV) for _ in range(ndir)]
dys = [torch.randn(T, B, H, generator=g).to(DEV) for _ in range(ndir)]
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def work():
    outs = []
    x = gxs
    for l in range(nl):
        flat = []
        for i in range(ndir):
            flat += [x[i], ws[i][0], ws[i][1], i == 1]
        ys = R.LSTMSeqGroupFn.apply(None, ndir, *flat)
        outs.append(ys)
        x = [torch.cat([ys[i]] * 4, dim=-1) * 0.5 + gxs[i] for i in range(ndir)]
    return outs
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ref = work()
    torch.cuda.synchronize()
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = work()
for r in range(4):
    gr.replay(); torch.cuda.synchronize()
    print("replay", r, "status", ops.lstm_seq_status(), [["%.2e" % float((out[l][i] - ref[l][i]).abs().max()) for i in range(ndir)] for l in range(nl)])
w = ops._lstm_seq_last_ws
print("ws ptr %x" % w.data_ptr(), "words:", w.view(torch.int32)[:8].tolist(), w.view(torch.int32)[60:72].tolist(), "flags sum", int(w.view(torch.int32)[64:320].sum()))
for l in range(nl):
    for i in range(ndir):
        print("y ptr %x" % out[l][i].data_ptr())
