import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from espnet_amd import ops
g = torch.Generator().manual_seed(21)
G, gb, T, H, D, L = 3, 10, 249, 4, 256, 6
kv = torch.randn(G * T, 2 * D * L, generator=g).cuda()
q = torch.randn(G * gb, D, generator=g).cuda()
outs = [ops.decode_src_attn(q, kv, 512, 768, 2 * D * L, None, G, gb, T, H).clone() for _ in range(20)]
torch.cuda.synchronize()
print("deterministic:", all(torch.equal(outs[0], o) for o in outs))
