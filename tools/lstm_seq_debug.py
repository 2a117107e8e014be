import sys, torch
sys.path.insert(0, '/root/repo')
import espnet_amd
from espnet_amd import ops, rnn_functional as R
DEV = torch.device('cuda')
def run(B, H, T, ndir, masked):
    g = torch.Generator().manual_seed(1)
    lens = sorted([max(1, T - (i * T) // (2 * B)) for i in range(B)], reverse=True) if masked else [T] * B
    live = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).to(torch.uint8).to(DEV).contiguous() if masked else None
    ws = [(torch.randn(4*H, H, generator=g).mul(1.0 / H ** 0.5).to(DEV), torch.randn(4*H, generator=g).mul(0.1).to(DEV)) for _ in range(ndir)]
    gxs = [torch.randn(T, B, 4*H, generator=g).to(DEV) for _ in range(ndir)]
    flat = []
    for i in range(ndir):
        flat += [gxs[i], ws[i][0], ws[i][1], i == 1]
    ys = R.LSTMSeqGroupFn.apply(live, ndir, *flat)
    print("status", ops.lstm_seq_status())
    for i in range(ndir):
        y2 = R.LSTMSeqFn.apply(gxs[i], ws[i][0], ws[i][1], live, i == 1)
        d = (ys[i] - y2).abs().amax(dim=(1, 2))
        print(B, H, T, ndir, masked, "dir", i, "max abs err per t:", [float("%.2e" % v) for v in d.tolist()])
        if d.max() > 1e-5:
            t = int((d > 1e-5).nonzero()[0])
            e = (ys[i][t] - y2[t]).abs()
            print("  first bad t", t, "bad rows", (e.amax(1) > 1e-5).nonzero().flatten().tolist()[:40], "bad cols count", int((e.amax(0) > 1e-5).sum()),
                  (e.amax(0) > 1e-5).nonzero().flatten().tolist()[:40])
for cfg in [(32, 1024, 6, 1, False), (32, 1024, 6, 2, False), (16, 320, 5, 2, False), (5, 64, 5, 2, True), (32, 1024, 6, 2, True)]:
    run(*cfg)
