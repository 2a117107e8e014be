"""GPU parity tests, kernel level: every HIP entry point against the CPU oracle / torch float64 on the
same seeded inputs, and against the vectors recorded from the reference (tests/golden).
Bar: integer work bit-exact; fp32 path rtol ~1e-4 (accumulation order differs from MKL);
bf16-MFMA path within the bf16 input-rounding bound (stated per test)."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, split_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from espnet_amd import ops as o
    o.set_precision("fp32")
    return o


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def report(name, a, b, tol):
    e = rel_err(a, b)
    mx = float((a.double().cpu() - b.double().cpu()).abs().max())
    print(f"[parity] {name}: rel_l2={e:.3e} max_abs={mx:.3e} (tol {tol:g})")
    assert e <= tol, f"{name}: rel err {e} > {tol}"


# ---------------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------------
GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (64, 64, 32), (65, 63, 33), (128, 128, 64), (257, 130, 100), (249, 249, 64),
               (300, 5000, 256), (7968 // 8, 256, 2048)]


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-6), ("bf16", 6e-3)])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_nt_shapes(ops, prec, tol, tile):
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in GEMM_SHAPES:
        A = torch.randn(M, K, generator=g)
        B = torch.randn(N, K, generator=g)
        bias = torch.randn(N, generator=g)
        C = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), C, M, N, K, K, K, N, bias=bias.to(DEV), tile=tile,
                 precision=0 if prec == "fp32" else 1)
        ref = A.double() @ B.double().t() + bias.double()
        report(f"gemm_nt[{prec},tile{tile}] {M}x{N}x{K}", C, ref, tol)


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-6), ("bf16", 6e-3)])
def test_gemm_layouts_batch_epilogues(ops, prec, tol):
    pr = 0 if prec == "fp32" else 1
    g = torch.Generator().manual_seed(1)
    M, N, K = 77, 45, 53
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    # NN (B stored [K,N])
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A.to(DEV), Bm.to(DEV), C, M, N, K, K, N, N, transB=1, precision=pr)
    report("gemm_nn", C, A.double() @ Bm.double(), tol)
    # TN (A stored [K,M]) + accumulate through beta
    At = torch.randn(K, M, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = C0.clone().to(DEV)
    ops.gemm(At.to(DEV), Bm.to(DEV), C, M, N, K, M, N, N, transA=1, transB=1, beta=1.0, alpha=0.5, precision=pr)
    report("gemm_tn_beta", C, 0.5 * (At.double().t() @ Bm.double()) + C0.double(), tol)
    # split-K atomics accumulate into C
    C = C0.clone().to(DEV)
    ops.gemm(At.to(DEV), Bm.to(DEV), C, M, N, K, M, N, N, transA=1, transB=1, splitk=3, precision=pr)
    report("gemm_tn_splitk", C, At.double().t() @ Bm.double() + C0.double(), tol)
    # epilogues: relu, swish, residual, prologue activation, masks
    Bn = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    aux = torch.randn(M, N, generator=g)
    base = A.double() @ Bn.double().t() + bias.double()
    for epi, fn in ((1, lambda v: v.clamp_min(0)), (2, lambda v: v * torch.sigmoid(v)),
                    (3, lambda v: v * (aux.double() > 0)),
                    (4, lambda v: v * (torch.sigmoid(aux.double()) * (1 + aux.double() * (1 - torch.sigmoid(aux.double())))))):
        C = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), Bn.to(DEV), C, M, N, K, K, K, N, bias=bias.to(DEV), epilogue=epi, aux=aux.to(DEV),
                 ldaux=N, R=R.to(DEV), ldr=N, alpha=0.7, precision=pr)
        report(f"gemm_epilogue{epi}", C, 0.7 * fn(base) + R.double(), tol)
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A.to(DEV), Bn.to(DEV), C, M, N, K, K, K, N, a_act=2, precision=pr)
    sw = A.double() * torch.sigmoid(A.double())
    report("gemm_a_act_swish", C, sw @ Bn.double().t(), tol)
    # two-level strided batch (attention layout): q,k [B,T,H,dk] -> scores [H,B,T1,ldp]
    Bb, T1, T2, H, dk = 3, 10, 13, 4, 16
    D = H * dk
    q = torch.randn(Bb, T1, D, generator=g)
    k = torch.randn(Bb, T2, D, generator=g)
    ldp = 16
    sc = torch.zeros(H * Bb * T1 * ldp, device=DEV)
    ops.gemm(q.to(DEV), k.to(DEV), sc, T1, T2, dk, D, D, ldp, batch=(Bb, H), sA=(T1 * D, dk), sB=(T2 * D, dk),
             sC=(T1 * ldp, Bb * T1 * ldp), precision=pr)
    ref = torch.einsum("bihd,bjhd->hbij", q.view(Bb, T1, H, dk).double(), k.view(Bb, T2, H, dk).double())
    report("gemm_batched_scores", sc.view(H, Bb, T1, ldp)[..., :T2], ref, tol)


@pytest.mark.parametrize("prec,tol", [("fp32", 3e-6), ("bf16", 6e-3)])
def test_gemm_implicit_conv(ops, prec, tol):
    """Conv2d(C,C,3,2) forward / weight-grad / input-grad as implicit GEMMs vs torch conv2d (float64)."""
    pr = 0 if prec == "fp32" else 1
    from espnet_amd import functional as F_
    g = torch.Generator().manual_seed(2)
    B, H1, W1, Cc = 2, 17, 9, 64
    H2, W2 = (H1 - 3) // 2 + 1, (W1 - 3) // 2 + 1
    y1 = torch.randn(B, H1, W1, Cc, generator=g)             # NHWC
    w = torch.randn(Cc, Cc, 3, 3, generator=g) * 0.1
    dy2 = torch.randn(B, H2, W2, Cc, generator=g)
    y1d = y1.permute(0, 3, 1, 2).double().requires_grad_(True)
    wd_ = w.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(y1d, wd_, stride=2)
    ref.backward(dy2.permute(0, 3, 1, 2).double())
    wf, wdd = ops.conv2_weight_prep(w.to(DEV))
    assert torch.equal(wf.cpu(), w.permute(2, 3, 1, 0).reshape(9, Cc, Cc))
    gth = ops.make_gather(Cc, F_._TAPS_FWD, H2, W2, H1, W1, 2, 2)
    M2 = B * H2 * W2
    y2 = torch.empty(M2, Cc, device=DEV)
    ops.gemm(y1.to(DEV), wf, y2, M2, Cc, 9 * Cc, 9 * Cc, Cc, Cc, transB=1, gather=gth, precision=pr)
    report("conv2_fwd", y2.view(B, H2, W2, Cc), ref.permute(0, 2, 3, 1), tol)
    dwf = torch.zeros(9 * Cc, Cc, device=DEV)
    ops.gemm(y1.to(DEV), dy2.reshape(M2, Cc).to(DEV), dwf, 9 * Cc, Cc, M2, 9 * Cc, Cc, Cc, transA=1, transB=1,
             gather=gth, splitk=2, tile=64, precision=pr)
    dw = torch.zeros(Cc, Cc, 3, 3, device=DEV)
    ops.conv2_weight_grad(dwf, dw, Cc, Cc)
    report("conv2_bwd_w", dw, wd_.grad, tol)
    dy1 = torch.full((B, H1, W1, Cc), float("nan"), device=DEV)
    ones = torch.ones(B, H1, W1, Cc, device=DEV)
    q0 = 0
    for (ph, pw), taps in F_._CLASSES:
        Ho, Wo = (H1 - ph + 1) // 2, (W1 - pw + 1) // 2
        gt = ops.make_gather(Cc, [((ph - kh) // 2, (pw - kw) // 2) for kh, kw in taps], Ho, Wo, H2, W2, 1, 1)
        cm = ops.make_rowmap(Ho, Wo, H1, W1, 2, ph, 2, pw)
        nt = len(taps)
        ops.gemm(dy2.reshape(M2, Cc).to(DEV), wdd, dy1, B * Ho * Wo, Cc, nt * Cc, nt * Cc, Cc, Cc, transB=1,
                 b_off=q0 * Cc * Cc, gather=gt, cmap=cm, epilogue=3, aux=ones, ldaux=Cc, precision=pr)
        q0 += nt
    report("conv2_bwd_x", dy1, y1d.grad.permute(0, 2, 3, 1), tol)


# ---------------------------------------------------------------------------------------------
# row kernels
# ---------------------------------------------------------------------------------------------
def test_layernorm_golden(ops):
    p, sd, grads = split_golden(load_golden("layernorm.npz"))
    x = p["x"].reshape(-1, 64).to(DEV)
    w, b = sd["weight"].to(DEV), sd["bias"].to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-12)
    report("layernorm_fwd", y, p["y"].reshape(-1, 64), 1e-6)
    dg, db = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    dx = ops.layernorm_bwd(p["gy"].reshape(-1, 64).to(DEV), x, w, mean, rstd, None, dg, db)
    report("layernorm_bwd_x", dx, p["gx"].reshape(-1, 64), 1e-5)
    report("layernorm_bwd_w", dg, grads["weight"], 1e-5)
    report("layernorm_bwd_b", db, grads["bias"], 1e-5)
    # d=256 vector path + residual-gradient fusion
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 256, generator=g)
    gy = torch.randn(1000, 256, generator=g)
    res = torch.randn(1000, 256, generator=g)
    w = torch.rand(256, generator=g) + 0.5
    b = torch.randn(256, generator=g)
    xd = x.double().requires_grad_(True)
    wd_, bd_ = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xd, (256,), wd_, bd_, 1e-12)
    yr.backward(gy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-12)
    report("layernorm256_fwd", y, yr, 1e-6)
    dg, db = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    dx = ops.layernorm_bwd(gy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, res.to(DEV), dg, db)
    report("layernorm256_bwd_x+res", dx, xd.grad + res.double(), 1e-5)
    report("layernorm256_bwd_w", dg, wd_.grad, 1e-5)
    report("layernorm256_bwd_b", db, bd_.grad, 1e-5)


def test_softmax_relshift_golden(ops, oracle):
    """rel_shift index map (bit-exact data movement) + masked softmax incl. a fully masked row."""
    p, _, _ = split_golden(load_golden("rel_shift.npz"))
    for key_x, key_y in (("x", "y"), ("x2", "y2")):
        x, want = p[key_x], p[key_y]
        b, h, t1, t2 = x.shape
        ld = (t2 + 3) // 4 * 4
        bd = torch.zeros(h * b * t1 * ld)
        bd.view(h, b, t1, ld)[..., :t2] = x.permute(1, 0, 2, 3)
        ac = torch.zeros_like(bd)
        P = torch.empty(h * b * t1 * ld, device=DEV)
        ops.softmax_fwd(ac.to(DEV), bd.to(DEV), None, P, h * b, b, t1, t2, ld, 1.0)
        ref = torch.softmax(want.double(), -1).permute(1, 0, 2, 3)
        report("softmax(rel_shift) " + key_x, P.view(h, b, t1, ld)[..., :t2], ref, 1e-6)
        assert float(P.view(h, b, t1, ld)[..., t2:].abs().sum()) == 0.0
    # masks + backward vs autograd (float64)
    g = torch.Generator().manual_seed(4)
    b, h, t1, t2 = 2, 3, 7, 7
    ld = 8
    ac = torch.randn(h, b, t1, ld, generator=g)
    bdm = torch.randn(h, b, t1, ld, generator=g)
    mask = torch.ones(b, t1, t2, dtype=torch.uint8)
    mask[1, :, 5:] = 0
    mask[0, 3, :] = 0     # fully masked query row
    dP = torch.randn(h, b, t1, ld, generator=g)
    acd = ac[..., :t2].double().requires_grad_(True)
    bdd = bdm[..., :t2].double().requires_grad_(True)
    sc = (acd + oracle.rel_shift(bdd.permute(1, 0, 2, 3)).permute(1, 0, 2, 3)) * 0.25
    mm = mask.bool().unsqueeze(0).eq(0)
    pr = torch.softmax(sc.masked_fill(mm, torch.finfo(torch.float64).min), -1).masked_fill(mm, 0.0)
    pr.backward(dP[..., :t2].double())
    P = torch.empty(h * b * t1 * ld, device=DEV)
    ops.softmax_fwd(ac.reshape(-1).to(DEV), bdm.reshape(-1).to(DEV), mask.to(DEV), P, h * b, b, t1, t2, ld, 0.25)
    report("masked_softmax_fwd", P.view(h, b, t1, ld)[..., :t2], pr, 1e-6)
    dS = dP.reshape(-1).clone().to(DEV)
    dbd = torch.full((h * b * t1 * ld,), float("nan"), device=DEV)      # the kernel defines every element itself
    ops.softmax_bwd(P, dS, dbd, h * b, t1, t2, ld, 0.25)
    report("masked_softmax_bwd_ac", dS.view(h, b, t1, ld)[..., :t2], acd.grad, 1e-5)
    report("masked_softmax_bwd_bd", dbd.view(h, b, t1, ld)[..., :t2], bdd.grad, 1e-5)
    # bf16-storage variant of the same kernels (P, dS, dbd in bf16): bf16 rounding of outputs only
    P16 = torch.empty(h * b * t1 * ld, device=DEV, dtype=torch.bfloat16)
    ops.softmax_fwd(ac.reshape(-1).to(DEV), bdm.reshape(-1).to(DEV), mask.to(DEV), P16, h * b, b, t1, t2, ld, 0.25)
    report("masked_softmax_fwd_bf16", P16.float().view(h, b, t1, ld)[..., :t2], pr, 4e-3)
    dS16 = torch.empty_like(P16)
    dbd16 = torch.full_like(P16, float("nan"))
    ops.softmax_bwd(P16, dP.reshape(-1).clone().to(DEV), dbd16, h * b, t1, t2, ld, 0.25, dS16=dS16)
    report("masked_softmax_bwd_ac_bf16", dS16.float().view(h, b, t1, ld)[..., :t2], acd.grad, 1e-2)
    report("masked_softmax_bwd_bd_bf16", dbd16.float().view(h, b, t1, ld)[..., :t2], bdd.grad, 1e-2)
    # register-resident vector form (bf16 storage, 1 / 2 / 4 chunks of 256 columns) against the fp32 kernels
    for (t1, t2, ld) in ((249, 249, 256), (300, 300, 304), (130, 600, 600)):
        b, h = 2, 2
        ac = torch.randn(h * b * t1 * ld, generator=g).to(DEV)
        bdm = torch.randn(h * b * t1 * ld, generator=g).to(DEV) if t1 == t2 else None
        mask = (torch.rand(b, 1, t2, generator=g) > 0.1).to(torch.uint8)
        mask[0, 0, :] = 1
        dP = torch.randn(h * b * t1 * ld, generator=g).to(DEV)
        P32 = torch.empty(h * b * t1 * ld, device=DEV)
        ops.softmax_fwd(ac, bdm, mask.to(DEV), P32, h * b, b, t1, t2, ld, 0.125)
        P16 = torch.empty(h * b * t1 * ld, device=DEV, dtype=torch.bfloat16)
        ops.softmax_fwd(ac, bdm, mask.to(DEV), P16, h * b, b, t1, t2, ld, 0.125)
        # same values up to the summation order of the row sum, rounded once: 1 bf16 ulp
        report("softmax_fwd vec %dx%d" % (t1, t2), P16.float(), P32, 3e-3)
        assert torch.equal(P16 == 0, P32.to(torch.bfloat16) == 0) or float(((P16 == 0) != (P32.to(torch.bfloat16) == 0)).sum()) < 8
        dS32, dbd32 = dP.clone(), torch.full((h * b * t1 * ld,), float("nan"), device=DEV)
        ops.softmax_bwd(P16.float(), dS32, dbd32 if bdm is not None else None, h * b, t1, t2, ld, 0.125)
        dS16 = torch.empty_like(P16)
        dbd16 = torch.full_like(P16, float("nan")) if bdm is not None else None
        ops.softmax_bwd(P16, dP.clone(), dbd16, h * b, t1, t2, ld, 0.125, dS16=dS16)
        v = lambda x: x.float().view(h * b, t1, ld)[..., :t2]                          # noqa: E731
        report("softmax_bwd vec %dx%d" % (t1, t2), v(dS16), v(dS32), 4e-3)
        assert float(dS16.float().view(h * b, t1, ld)[..., t2:].abs().sum()) == 0.0
        if bdm is not None:
            report("softmax_bwd vec dbd %dx%d" % (t1, t2), dbd16.float(), dbd32, 4e-3)


def test_lsm_loss_golden(ops):
    p, _, _ = split_golden(load_golden("lsm_loss.npz"))
    x = p["x"].reshape(-1, 17).to(DEV)
    t = p["t"].reshape(-1).to(DEV)
    rows, correct, grad = ops.lsm_loss(x, t, 0.1, 1.0 / 3, -1)
    report("lsm_loss", ops.reduce_sum(rows, 1.0 / 3), p["loss_n0"], 2e-6)
    report("lsm_grad", grad.view(3, 5, 17), p["gx_n0"], 1e-5)
    n = int((p["t"] != -1).sum())
    rows, _, grad = ops.lsm_loss(x, t, 0.1, 1.0 / n, -1)
    report("lsm_loss_lengthnorm", ops.reduce_sum(rows, 1.0 / n), p["loss_n1"], 2e-6)
    report("lsm_grad_lengthnorm", grad.view(3, 5, 17), p["gx_n1"], 1e-5)
    rows, _, _ = ops.lsm_loss(x, t, 0.0, 1.0 / 3, -1, want_grad=False)
    report("lsm_loss_smoothing0", ops.reduce_sum(rows, 1.0 / 3), p["loss_s0"], 2e-6)
    acc = float(correct.sum()) / n
    assert abs(acc - float(p["acc"])) < 1e-7


def test_argmax_and_collapse_bit_exact(ops, oracle):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 5000, generator=g)
    x[7, 100] = x[7, 4000] = 50.0      # tie -> lowest index
    got = ops.argmax_rows(x.to(DEV)).cpu()
    assert torch.equal(got.long(), x.argmax(-1))
    ids = torch.randint(0, 4, (5, 40), generator=g, dtype=torch.int32)
    hl = torch.tensor([40, 33, 1, 0, 17], dtype=torch.int32)
    out, n = ops.ctc_collapse(ids.to(DEV), hl.to(DEV), 0)
    for b in range(5):
        row = ids[b, : int(hl[b])]
        want, prev = [], None
        for v in row.tolist():
            if v != prev:
                if v != 0:
                    want.append(v)
                prev = v
        assert out[b, : int(n[b])].tolist() == want
        assert (out[b, int(n[b]):] == -1).all()


def test_add_sos_eos_bit_exact(ops, oracle):
    ys = torch.tensor([[3, 4, 5, -1], [7, -1, -1, -1], [1, 2, 3, 4], [-1, -1, -1, -1]])
    a, b, n = ops.add_sos_eos(ys.to(DEV), 9, 9, -1)
    ra, rb = oracle.add_sos_eos(ys, 9, 9, -1)
    assert torch.equal(a.cpu(), ra) and torch.equal(b.cpu(), rb)
    assert n.tolist() == [3, 1, 4, 0]


def test_ctc_golden(ops):
    p, _, _ = split_golden(load_golden("ctc.npz"))
    lg = p["logits"].contiguous().to(DEV)
    hl = p["hlens"].to(torch.int32).to(DEV)
    nll, grad = ops.ctc_loss(lg, p["ys"].to(DEV), hl, 0, -1, 1.0 / 3)
    report("ctc_nll", nll, p["nll"], 2e-6)
    report("ctc_loss", ops.reduce_sum(nll, 1.0 / 3), p["loss"], 2e-6)
    report("ctc_grad", grad, p["glogits"], 2e-5)
    assert float(grad[1, 4].abs().sum()) == 0.0          # frames beyond hlens get zero gradient
    # infeasible alignment -> +inf, never an error code
    nll, _ = ops.ctc_loss(lg[:1, :3].contiguous(), torch.tensor([[3, 3, 4]]).to(DEV),
                          torch.tensor([3], dtype=torch.int32).to(DEV), 0, -1, 1.0, want_grad=False)
    assert math.isinf(float(nll[0])) and float(nll[0]) > 0


def test_ctc_random_vs_oracle(ops, oracle):
    """ragged lengths, repeated labels, empty label sequence; loss and gradient vs torch CPU"""
    g = torch.Generator().manual_seed(6)
    B, T, V, L = 6, 50, 40, 12
    x = torch.randn(B, T, V, generator=g)
    ys = torch.randint(1, V, (B, L), generator=g)
    ys[0, :] = 5                      # all-repeated labels
    ys[1, 3:] = -1
    ys[2, :] = -1                     # empty
    ys[3, 1:] = ys[3, 0:1]
    hl = torch.tensor([50, 40, 10, 50, 25, 49], dtype=torch.int32)
    xd = x.double().requires_grad_(True)
    ref = oracle.ctc_loss(xd, hl, ys)
    ref.backward()
    nll, grad = ops.ctc_loss(x.to(DEV), ys.to(DEV), hl.to(DEV), 0, -1, 1.0 / B)
    report("ctc_random_loss", ops.reduce_sum(nll, 1.0 / B), ref, 1e-5)
    report("ctc_random_grad", grad, xd.grad, 1e-4)
    # property: every valid frame's gradient row sums to zero (softmax minus occupancies)
    rs = grad.sum(-1).abs().max()
    assert float(rs) < 1e-5


# ---------------------------------------------------------------------------------------------
# element-wise / conv module pieces
# ---------------------------------------------------------------------------------------------
def test_elementwise(ops):
    g = torch.Generator().manual_seed(7)
    a = torch.randn(33, 2 * 48, generator=g)
    dy = torch.randn(33, 48, generator=g)
    ad = a.double().requires_grad_(True)
    yr = torch.nn.functional.glu(ad, -1)
    yr.backward(dy.double())
    report("glu_fwd", ops.glu_fwd(a.to(DEV), 48), yr, 1e-6)
    report("glu_bwd", ops.glu_bwd(dy.to(DEV), a.to(DEV), 48), ad.grad, 1e-6)
    x = torch.randn(1001, generator=g)
    y = torch.randn(1001, generator=g)
    report("axpby", ops.axpby(x.to(DEV), y.to(DEV), 0.3, -1.5), 0.3 * x.double() - 1.5 * y.double(), 1e-6)
    m = torch.randn(500, 70, generator=g)
    out = torch.ones(70, device=DEV)
    ops.colsum(m.to(DEV), out, 2.0)
    report("colsum", out, 1 + 2 * m.double().sum(0), 1e-5)
    tok = torch.randint(0, 11, (3, 5), generator=g)
    table = torch.randn(11, 16, generator=g)
    pe = torch.randn(9, 16, generator=g)
    e = ops.embed_pe(tok.to(DEV), table.to(DEV), pe.to(DEV), 5, 4.0)
    report("embed_pe", e.view(3, 5, 16), table[tok].double() * 4 + pe[:5].double(), 1e-6)
    dt = torch.zeros(11, 16, device=DEV)
    do = torch.randn(15, 16, generator=g)
    ops.embed_bwd(tok.reshape(-1).to(DEV), do.to(DEV), dt, 4.0)
    ref = torch.zeros(11, 16, dtype=torch.float64).index_add_(0, tok.reshape(-1), do.double() * 4)
    report("embed_bwd", dt, ref, 1e-5)


def test_dwconv_bn(ops):
    g = torch.Generator().manual_seed(8)
    B, T, Cc, K = 3, 29, 64, 15
    x = torch.randn(B, T, Cc, generator=g)
    w = torch.randn(Cc, 1, K, generator=g)
    bias = torch.randn(Cc, generator=g)
    dy = torch.randn(B, T, Cc, generator=g)
    xd = x.double().requires_grad_(True)
    wd_, bd_ = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xd.transpose(1, 2), wd_, bd_, padding=(K - 1) // 2, groups=Cc).transpose(1, 2)
    yr.backward(dy.double())
    y = ops.dwconv_fwd(x.to(DEV), w.view(Cc, K).to(DEV), bias.to(DEV), B, T, Cc, K)
    report("dwconv_fwd", y, yr, 1e-6)
    report("dwconv_bwd_x", ops.dwconv_bwd_x(dy.to(DEV), w.view(Cc, K).to(DEV), B, T, Cc, K), xd.grad, 1e-6)
    dw, db = torch.zeros(Cc, K, device=DEV), torch.zeros(Cc, device=DEV)
    ops.dwconv_bwd_w(dy.to(DEV), x.to(DEV), dw, db, B, T, Cc, K)
    report("dwconv_bwd_w", dw, wd_.grad.view(Cc, K), 1e-5)
    report("dwconv_bwd_b", db, bd_.grad, 1e-5)
    # batch norm (train) + swish
    M = B * T
    xb = (torch.randn(M, Cc, generator=g) * 2 + 1)
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    xbd = xb.double().requires_grad_(True)
    gd, bd2 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    z = torch.nn.functional.batch_norm(xbd, rmd, rvd, gd, bd2, True, 0.1, 1e-5)
    yr = z * torch.sigmoid(z)
    dyb = torch.randn(M, Cc, generator=g)
    yr.backward(dyb.double())
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    nbt = torch.full((), 7, dtype=torch.int64, device=DEV)
    mean, rstd = ops.bn_stats(xb.to(DEV), M, Cc, 1e-5, 0.1, rmg, rvg, nbt)
    assert int(nbt) == 8          # num_batches_tracked += 1 in the statistics launch
    y = ops.bn_apply(xb.to(DEV), mean, rstd, gam.to(DEV), bet.to(DEV), M, Cc, 2)
    report("bn_swish_fwd", y, yr, 2e-6)
    report("bn_running_mean", rmg, rmd, 1e-6)
    report("bn_running_var", rvg, rvd, 1e-6)
    dg, dbt = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    dx = ops.bn_bwd(dyb.to(DEV), xb.to(DEV), mean, rstd, gam.to(DEV), bet.to(DEV), dg, dbt, M, Cc, 2, True)
    report("bn_swish_bwd_x", dx, xbd.grad, 2e-5)
    report("bn_bwd_gamma", dg, gd.grad, 2e-5)
    report("bn_bwd_beta", dbt, bd2.grad, 2e-5)


def test_conv1(ops):
    g = torch.Generator().manual_seed(9)
    B, T, F, Cc = 2, 21, 20, 64
    x = torch.randn(B, T, F, generator=g)
    w = torch.randn(Cc, 1, 3, 3, generator=g)
    b = torch.randn(Cc, generator=g)
    wd_, bd_ = w.double().requires_grad_(True), b.double().requires_grad_(True)
    pre = torch.nn.functional.conv2d(x.double().unsqueeze(1), wd_, bd_, stride=2)
    yr = torch.relu(pre)
    y = ops.conv1_fwd(x.to(DEV), w.to(DEV), b.to(DEV), B, T, F, Cc)
    report("conv1_fwd", y, yr.permute(0, 2, 3, 1), 1e-6)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    dyn = (dy * (pre > 0)).permute(0, 2, 3, 1).contiguous().float()
    dw, db = torch.zeros(Cc, 9, device=DEV), torch.zeros(Cc, device=DEV)
    ops.conv1_bwd_w(dyn.to(DEV), x.to(DEV), dw, db, B, T, F, Cc)
    report("conv1_bwd_w", dw, wd_.grad.view(Cc, 9), 1e-5)
    report("conv1_bwd_b", db, bd_.grad, 1e-5)
    # bf16 activations (two channels per thread, packed stores; bf16 gradient input) and a ragged width (W = 41 > 40)
    for (B2, T2, F2, C2) in ((2, 21, 20, 64), (3, 17, 84, 128)):
        x2 = torch.randn(B2, T2, F2, generator=g)
        w2 = torch.randn(C2, 1, 3, 3, generator=g)
        b2 = torch.randn(C2, generator=g)
        ref = torch.relu(torch.nn.functional.conv2d(x2.double().unsqueeze(1), w2.double(), b2.double(), stride=2))
        ref = ref.permute(0, 2, 3, 1)
        y32 = ops.conv1_fwd(x2.to(DEV), w2.to(DEV), b2.to(DEV), B2, T2, F2, C2)
        report("conv1_fwd %dx%d" % (F2, C2), y32, ref, 1e-6)
        y16 = ops.conv1_fwd(x2.to(DEV), w2.to(DEV), b2.to(DEV), B2, T2, F2, C2, torch.bfloat16)
        assert torch.equal(y16, y32.to(torch.bfloat16))                 # same values, rounded once
        dy2 = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
        dwr = torch.einsum("bhwc,bhwk->ck", dy2.double(),
                           torch.nn.functional.unfold(x2.double().unsqueeze(1), 3, stride=2)
                           .transpose(1, 2).reshape(B2, ref.shape[1], ref.shape[2], 9))
        dw2, db2 = torch.zeros(C2, 9, device=DEV), torch.zeros(C2, device=DEV)
        ops.conv1_bwd_w(dy2.to(DEV), x2.to(DEV), dw2, db2, B2, T2, F2, C2)
        report("conv1_bwd_w bf16 %dx%d" % (F2, C2), dw2, dwr, 1e-5)
        report("conv1_bwd_b bf16 %dx%d" % (F2, C2), db2, dy2.double().sum((0, 1, 2)), 1e-5)


@pytest.mark.parametrize("shape", [(2, 37, 20, 64), (2, 29, 20, 64), (3, 29, 80, 16), (4, 100, 80, 256), (32, 1000, 80, 256)])
@pytest.mark.parametrize("bf16", [False, True])
def test_conv1_bwd_w_deterministic(ops, shape, bf16):
    """eamd_conv1_bwd_w against the float64 weight gradient of conv2d at the shapes of the Conv2dSubsampling fixtures
    (few row blocks), a mid-size batch (hundreds of row blocks) and config 2 itself, with fp32 (one channel per thread)
    and bf16 (two channels per thread) gradients; launched twice into the same zeroed buffers: the two results must
    be bit-identical (the per-block partial sums are reduced in a fixed order; nothing is added with float atomics
    except the final <= 16 slice sums per element) and accumulate (dw += ...)."""
    B, T, F, Cc = shape
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, T, F, generator=g)
    H, W = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    dy = torch.randn(B, H, W, Cc, generator=g)
    dy = dy * (torch.rand(B, H, W, Cc, generator=g) > 0.5)          # ReLU-masked, as the caller hands it in
    if bf16:
        dy = dy.to(torch.bfloat16)
    cols = torch.nn.functional.unfold(x.double().unsqueeze(1), 3, stride=2).transpose(1, 2).reshape(B, H, W, 9)
    dwr = torch.einsum("bhwc,bhwk->ck", dy.double(), cols)
    dbr = dy.double().sum((0, 1, 2))
    outs = []
    for _ in range(2):
        dw, db = torch.zeros(Cc, 9, device=DEV), torch.zeros(Cc, device=DEV)
        ops.conv1_bwd_w(dy.to(DEV), x.to(DEV), dw, db, B, T, F, Cc)
        outs.append((dw.clone(), db.clone()))
    report("conv1_bwd_w %s bf16=%s" % (shape, bf16), outs[0][0], dwr, 2e-5)
    report("conv1_bwd_b %s bf16=%s" % (shape, bf16), outs[0][1], dbr, 2e-5)
    if B * H <= 64:          # at most two reduction slices (a + b == b + a): bit-reproducible
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ops.conv1_bwd_w(dy.to(DEV), x.to(DEV), dw, db, B, T, F, Cc)          # accumulates into non-zero buffers
    report("conv1_bwd_w accumulate %s" % (shape,), dw, 2 * dwr, 2e-5)


def test_optimizer(ops):
    """Adam + Noam schedule + clipping vs torch.optim.Adam / clip_grad_norm_ (fp32 CPU)."""
    g = torch.Generator().manual_seed(10)
    n = 10007
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    pad = (4 - n % 4) % 4
    p = torch.zeros(n + pad)
    p[:n] = p0
    p, m, v = p.to(DEV), torch.zeros(n + pad, device=DEV), torch.zeros(n + pad, device=DEV)
    state = torch.zeros(8, device=DEV)
    ws = torch.empty(1024, device=DEV)
    gn = torch.empty(1, device=DEV)
    factor, dmodel, warm = 1.0, 256.0, 25.0
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * (10.0 if step == 3 else 0.1)
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], 5.0)
        lr = factor * dmodel ** -0.5 * min(step ** -0.5, step * warm ** -1.5)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        gfull = torch.zeros(n + pad)
        gfull[:n] = gr
        gd = gfull.to(DEV)
        ops.grad_norm(gd, ws, gn)
        ops.sched_step(state, gn, 1, 0.0, factor, dmodel, warm, 0.9, 0.98, 5.0)
        ops.adam_step(p, gd, m, v, state, 0.9, 0.98, 1e-9, 0.0)
        assert abs(float(gn) - float(gr.norm())) <= 1e-5 * float(gr.norm())
        assert abs(float(state[1]) - lr) <= 1e-6 * lr
    report("adam_5_steps", p[:n], pr.detach(), 1e-5)
    # non-finite gradient -> step skipped, parameters untouched (trainer.py:439-455)
    before = p.clone()
    gd = torch.full((n + pad,), float("nan"), device=DEV)
    ops.grad_norm(gd, ws, gn)
    ops.sched_step(state, gn, 1, 0.0, factor, dmodel, warm, 0.9, 0.98, 5.0)
    ops.adam_step(p, gd, m, v, state, 0.9, 0.98, 1e-9, 0.0)
    assert torch.equal(p, before) and float(state[0]) == 5.0 and float(state[5]) == 1.0


def test_adadelta_and_warmuplr_golden(ops):
    """eamd_adadelta_step (+ the trainer's clipping and eps decay) against the trajectory torch.optim.Adadelta produced
    inside the reference's own loop pieces (asr.py:505-508, asr_utils.py:517-528), and eamd_sched_step mode 2 against
    the learning rates espnet2's WarmupLR (schedulers/warmup_lr.py:10-53) handed to 14 optimizer steps"""
    from conftest import load_golden
    from espnet_amd import train
    g = load_golden("adadelta.npz")
    p0, grads, traj = torch.from_numpy(g["p0"]), torch.from_numpy(g["grads"]), torch.from_numpy(g["traj"])
    lin = torch.nn.Linear(p0.numel(), 1, bias=False).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(p0.view(1, -1))
    flat = train.FlatParams(lin)
    opt = train.Adadelta(flat, lr=1.0, rho=0.95, eps=1e-8, weight_decay=0.0, max_grad_norm=5.0)
    for step in range(grads.shape[0]):
        flat.grad.zero_()
        lin.weight._eamd_grad.copy_(grads[step].view(1, -1))
        opt.step()
        report("adadelta step %d" % step, lin.weight.detach().view(-1), traj[step], 2e-6)
        if step == 2:
            opt.eps_decay(0.01)
    assert abs(opt.eps - float(g["eps_after"])) < 1e-12 * 1e-8 + 1e-16
    assert opt.stats()["step"] == grads.shape[0]
    # non-finite gradient: skipped
    before = lin.weight.detach().clone()
    lin.weight._eamd_grad.fill_(float("inf"))
    opt.step()
    assert torch.equal(lin.weight.detach(), before) and opt.stats()["skipped"] == 1
    w = load_golden("warmup_lr.npz")
    state, gn = torch.zeros(8, device=DEV), torch.ones(1, device=DEV)
    for k, want in enumerate(w["lrs"].tolist()):
        ops.sched_step(state, gn, 2, float(w["base_lr"]), 1.0, 1.0, float(w["warmup"]), 0.9, 0.98, 0.0)
        assert abs(float(state[1]) - want) <= 2e-7 * want, (k, float(state[1]), want)


# ---------------------------------------------------------------------------------------------
# bf16-operand fast GEMM (gemm_bf16.hip): inputs are rounded to bf16 on the host first, so the only
# difference to the float64 reference is fp32 accumulation order -> tight tolerance, which pins the
# ds_read_b64_tr_b16 fragment maps of the transposed layouts exactly (asymmetric random operands).
# ---------------------------------------------------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("tile", [0, 64, 128])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16_layouts(ops, tile, ta, tb):
    g = torch.Generator().manual_seed(11)
    for (M, N, K) in [(16, 16, 32), (64, 64, 64), (77, 45, 53), (130, 257, 100), (256, 2048, 996), (249, 64, 249),
                      (128, 128, 8)]:
        A = _bf(torch.randn(K, M, generator=g) if ta else torch.randn(M, K, generator=g))
        B = _bf(torch.randn(K, N, generator=g) if tb else torch.randn(N, K, generator=g))
        Ad = (A.double().t() if ta else A.double())
        Bd = (B.double() if tb else B.double().t())
        ref = Ad @ Bd
        C = torch.empty(M, N, device=DEV)
        Cb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(A.to(DEV), B.to(DEV), C, M, N, K, M if ta else K, N if tb else K, N, transA=ta, transB=tb,
                 tile=tile, Cb=Cb)
        report(f"gemm_bf16[ta{ta},tb{tb},tile{tile}] {M}x{N}x{K}", C, ref, 2e-6)
        report(f"gemm_bf16 Cb copy", Cb.float(), ref, 4e-3)


def test_gemm_bf16_epilogue_batch_gather(ops):
    from espnet_amd import functional as F_
    g = torch.Generator().manual_seed(12)
    M, N, K = 90, 70, 130
    A, Bn = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g))
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    aux = _bf(torch.randn(M, N, generator=g))
    base = A.double() @ Bn.double().t() + bias.double()
    sg = torch.sigmoid(aux.double())
    for epi, fn in ((1, lambda v: v.clamp_min(0)), (2, lambda v: v * torch.sigmoid(v)),
                    (3, lambda v: v * (aux.double() > 0)), (4, lambda v: v * (sg * (1 + aux.double() * (1 - sg))))):
        C = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), Bn.to(DEV), C, M, N, K, K, K, N, bias=bias.to(DEV), epilogue=epi, aux=aux.to(DEV), ldaux=N,
                 R=R.to(DEV), ldr=N, alpha=0.7)
        report(f"gemm_bf16_epilogue{epi}", C, 0.7 * fn(base) + R.double(), 3e-6)
    # prologue activation (rounded back to bf16 before the MFMA) + split-K + fused column sums
    At = _bf(torch.randn(K, M, generator=g))
    Bt = _bf(torch.randn(K, N, generator=g))
    C0 = torch.randn(M, N, generator=g)
    C = C0.clone().to(DEV)
    cs = torch.ones(M, device=DEV)
    ops.gemm(At.to(DEV), Bt.to(DEV), C, M, N, K, M, N, N, transA=1, transB=1, splitk=3, alpha=0.5, colsum=cs)
    report("gemm_bf16_tn_splitk", C, 0.5 * (At.double().t() @ Bt.double()) + C0.double(), 3e-6)
    report("gemm_bf16_colsum", cs, 1 + 0.5 * At.double().sum(0), 1e-5)
    sw = _bf((Bt.float() * torch.sigmoid(Bt.float()))).double()
    C = torch.empty(M, N, device=DEV)
    ops.gemm(At.to(DEV), Bt.to(DEV), C, M, N, K, M, N, N, transA=1, transB=1, b_act=2)
    report("gemm_bf16_b_act_swish", C, At.double().t() @ sw, 2e-3)
    # attention layout, both products, through transposed reads
    Bb, T1, T2, H, dk = 3, 10, 13, 4, 16
    D = H * dk
    q, k, v = (_bf(torch.randn(Bb, T, D, generator=g)) for T in (T1, T2, T2))
    ldp = 16
    sc = torch.zeros(H * Bb * T1 * ldp, device=DEV)
    ops.gemm(q.to(DEV), k.to(DEV), sc, T1, T2, dk, D, D, ldp, batch=(Bb, H), sA=(T1 * D, dk), sB=(T2 * D, dk),
             sC=(T1 * ldp, Bb * T1 * ldp))
    ref = torch.einsum("bihd,bjhd->hbij", q.view(Bb, T1, H, dk).double(), k.view(Bb, T2, H, dk).double())
    report("gemm_bf16_scores", sc.view(H, Bb, T1, ldp)[..., :T2], ref, 3e-6)
    P = _bf(torch.randn(H, Bb, T1, ldp, generator=g))
    P[..., T2:] = 0
    cx = torch.empty(Bb * T1, D, device=DEV)
    ops.gemm(P.to(DEV), v.to(DEV), cx, T1, dk, T2, ldp, D, D, transB=1, batch=(Bb, H), sA=(T1 * ldp, Bb * T1 * ldp),
             sB=(T2 * D, dk), sC=(T1 * D, dk))
    ref = torch.einsum("hbij,bjhd->bihd", P[..., :T2].double(), v.view(Bb, T2, H, dk).double()).reshape(Bb * T1, D)
    report("gemm_bf16_context", cx, ref, 3e-6)
    # implicit conv (forward / weight-grad / input-grad) with bf16 activations
    B_, H1, W1, Cc = 2, 17, 9, 64
    H2, W2 = (H1 - 3) // 2 + 1, (W1 - 3) // 2 + 1
    y1 = _bf(torch.randn(B_, H1, W1, Cc, generator=g))
    w = _bf(torch.randn(Cc, Cc, 3, 3, generator=g) * 0.1)
    dy2 = _bf(torch.randn(B_, H2, W2, Cc, generator=g))
    y1d = y1.permute(0, 3, 1, 2).double().requires_grad_(True)
    wdd_ = w.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(y1d, wdd_, stride=2)
    ref.backward(dy2.permute(0, 3, 1, 2).double())
    wf = w.permute(2, 3, 1, 0).reshape(9, Cc, Cc).contiguous().to(DEV)
    order = [0, 2, 6, 8, 1, 7, 3, 5, 4]
    wd = w.permute(2, 3, 0, 1).reshape(9, Cc, Cc)[order].contiguous().to(DEV)
    gth = ops.make_gather(Cc, F_._TAPS_FWD, H2, W2, H1, W1, 2, 2)
    M2 = B_ * H2 * W2
    y2 = torch.empty(M2, Cc, device=DEV)
    ops.gemm(y1.to(DEV), wf, y2, M2, Cc, 9 * Cc, 9 * Cc, Cc, Cc, transB=1, gather=gth)
    report("conv2_bf16_fwd", y2.view(B_, H2, W2, Cc), ref.permute(0, 2, 3, 1), 3e-6)
    dwf = torch.zeros(9 * Cc, Cc, device=DEV)
    ops.gemm(y1.to(DEV), dy2.reshape(M2, Cc).to(DEV), dwf, 9 * Cc, Cc, M2, 9 * Cc, Cc, Cc, transA=1, transB=1,
             gather=gth, splitk=2, tile=64)
    dw = torch.zeros(Cc, Cc, 3, 3, device=DEV)
    ops.conv2_weight_grad(dwf, dw, Cc, Cc)
    report("conv2_bf16_bwd_w", dw, wdd_.grad, 3e-6)
    dy1 = torch.full((B_, H1, W1, Cc), float("nan"), device=DEV)
    ones = torch.ones(B_, H1, W1, Cc, device=DEV, dtype=torch.bfloat16)
    q0 = 0
    for (ph, pw), taps in F_._CLASSES:
        Ho, Wo = (H1 - ph + 1) // 2, (W1 - pw + 1) // 2
        gt = ops.make_gather(Cc, [((ph - kh) // 2, (pw - kw) // 2) for kh, kw in taps], Ho, Wo, H2, W2, 1, 1)
        cm = ops.make_rowmap(Ho, Wo, H1, W1, 2, ph, 2, pw)
        nt = len(taps)
        ops.gemm(dy2.reshape(M2, Cc).to(DEV), wd, dy1, B_ * Ho * Wo, Cc, nt * Cc, nt * Cc, Cc, Cc, transB=1,
                 b_off=q0 * Cc * Cc, gather=gt, cmap=cm, epilogue=3, aux=ones, ldaux=Cc)
        q0 += nt
    report("conv2_bf16_bwd_x", dy1, y1d.grad.permute(0, 2, 3, 1), 3e-6)


def test_gemm_fused_dropout_matches_dropout_kernel(ops):
    """GEMM epilogue dropout draws the very mask eamd_dropout draws for the same (step, salt, index)"""
    import espnet_amd
    espnet_amd.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(11)
        M, N, K = 300, 192, 128
        x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
        W = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
        b = torch.randn(N, generator=g).to(DEV)
        R = torch.randn(M, N, generator=g).to(DEV)
        p, salt = 0.3, 77
        plain = ops.linear_fwd(x, W, b)
        fused = ops.linear_fwd(x, W, b, R=R, alpha=0.5, drop=(p, salt))
        want = R + 0.5 * ops.dropout(plain, p, salt)
        assert torch.equal(fused == R, want == R)                      # identical keep pattern
        report("gemm fused dropout", fused, want, 1e-6)
        kept = float((fused != R).float().mean())
        assert abs(kept - (1 - p)) < 0.02
        # dual output: z untouched, h = dropout(swish(z))
        z = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        h = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.linear_fwd(x, W, b, out=z, drop=(p, salt), Hb=h, h_act=ops.ACT_SWISH)
        report("gemm dual z", z.float(), plain, 4e-3)
        hw = ops.dropout(plain, p, salt, act=ops.ACT_SWISH)
        assert torch.equal(h.float() == 0, hw == 0) or float(((h.float() == 0) != (hw == 0)).float().mean()) < 1e-4
        report("gemm dual h", h.float(), hw, 4e-3)
        # same masks on the backward side: dz = drop(dswish(z) * (dy W))
        dy = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
        Wt = torch.randn(K, N, generator=g).to(DEV).to(torch.bfloat16)
        a = ops.linear_bwd_x(dy, Wt, drop=(p, salt))
        bb = ops.dropout(ops.linear_bwd_x(dy, Wt), p, salt)
        report("gemm bwd_x fused dropout", a, bb, 1e-6)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("shape", [(300, 192, 128), (7968, 256, 2048), (7968, 256, 256)])
def test_gemm_f32_fused_dropout_matches_dropout_kernel(ops, shape):
    """fp32 mode: the dropout epilogue of the pipelined fp32 GEMM draws the very mask eamd_dropout draws for the same
    (step, salt, element index), for both tile sizes, with bias, alpha and the residual behind it"""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + K)
    x, W = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV)
    b, R = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    p, salt = 0.1, 4242
    plain = ops.linear_fwd(x, W, b)
    fused = ops.linear_fwd(x, W, b, R=R, alpha=0.5, drop=(p, salt))
    want = R + 0.5 * ops.dropout(plain, p, salt)
    assert torch.equal(fused == R, want == R)                          # identical keep pattern
    report("fp32 gemm fused dropout %s" % (shape,), fused, want, 1e-6)
    assert abs(float((fused != R).float().mean()) - (1 - p)) < 0.01
    report("fp32 gemm vs float64 %s" % (shape,), plain, x.double().cpu() @ W.double().cpu().t() + b.double().cpu(), 2e-6)
    # second fp32 output (h_dtype 1): z untouched, h = dropout(swish(z)) from the same launch (FFN up-projection)
    from espnet_amd.ops import ACT_SWISH
    h = torch.empty(M, N, device=DEV)
    z = ops.linear_fwd(x, W, b, drop=(p, salt), Hb=h, h_act=ACT_SWISH)
    assert torch.equal(z, plain)
    report("fp32 gemm dual output h %s" % (shape,), h, ops.dropout(plain, p, salt, act=ACT_SWISH), 1e-6)


@pytest.mark.parametrize("shape", [(300, 192, 128), (7968, 256, 2048), (1000, 2048, 256)])
def test_gemm_f32_operand_dropout_matches_dropout_kernel(ops, shape):
    """fp32 mode: dropout applied to a GEMM operand while it is staged (eamd_gemm_t.a_drop_p / b_drop_p) gives what
    the product of the materialised eamd_dropout result gives - forward (A = drop(act(z))), input gradient
    (A = drop(dy)) and weight gradient (A = drop(dy), B = drop(act(z)), bias gradient = column sums of drop(dy))"""
    from espnet_amd.ops import ACT_SWISH, ACT_NONE
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * K)
    z, W = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b, dy = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    p_in, s_in, p_out, s_out = 0.1, 777, 0.2, 991
    h = ops.dropout(z, p_in, s_in, act=ACT_SWISH)
    report("operand drop fwd %s" % (shape,), ops.linear_fwd(z, W, b, a_act=ACT_SWISH, a_drop=(p_in, s_in)),
           ops.linear_fwd(h, W, b), 2e-6)
    dyd = ops.dropout(dy, p_out, s_out)
    report("operand drop bwd_x %s" % (shape,), ops.linear_bwd_x(dy, W, a_drop=(p_out, s_out)), ops.linear_bwd_x(dyd, W), 2e-6)
    # with the epilogue dropout behind it (the FFN's dz)
    report("operand drop bwd_x + epilogue drop %s" % (shape,),
           ops.linear_bwd_x(dy, W, a_drop=(p_out, s_out), drop=(p_in, s_in)),
           ops.dropout(ops.linear_bwd_x(dyd, W), p_in, s_in), 2e-6)
    dW1, db1 = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    dW2, db2 = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    ops.linear_bwd_w(dy, z, dW1, alpha=0.5, b_act=ACT_SWISH, db=db1, a_drop=(p_out, s_out), b_drop=(p_in, s_in))
    ops.linear_bwd_w(dyd, h, dW2, alpha=0.5, b_act=ACT_NONE, db=db2)
    torch.cuda.synchronize()
    report("operand drop bwd_w %s" % (shape,), dW1, dW2, 2e-5)
    report("operand drop bias grad %s" % (shape,), db1, db2, 2e-5)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_grouped_weight_gradient_launch_vs_float64(ops, prec):
    """eamd_gemm_group_plan / eamd_gemm_group_launch through ops.linear_bwd_w's queue: a mix of weight-gradient problems
    (square, skinny, ragged edges in every dimension, with and without bias-gradient column sums, alpha != 1, split-K 1
    and > 1; in fp32 mode large ones on the 128x128 grouped launch) accumulated ON TOP of existing arena contents -
    against float64; and a queue of one falls back to a plain launch"""
    import espnet_amd
    espnet_amd.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(3)
        adt = ops.act_dtype()
        shapes = [(7968, 256, 256, True, 1.0), (3000, 768, 256, True, 0.5), (1000, 50, 256, True, 1.0), (517, 130, 70, False, 1.0),
                  (7968, 2048, 256, True, 1.0), (4000, 256, 2048, False, 2.0), (96, 64, 64, True, 1.0)]
        for sel in (shapes, shapes[:1]):
            up8 = lambda v: (v + 7) // 8 * 8  # noqa: E731
            total = sum(up8(n * k) + up8(n) for _m, n, k, _b, _a in sel)
            arena = torch.randn(total, generator=g).to(DEV)
            before = arena.clone()
            ops.register_grad_arena(arena)
            ops_in, off = [], 0
            rec = []
            ops._gemm_record = rec
            ops.wgrad_group_begin()
            for m, n, k, has_b, alpha in sel:
                dy = (0.1 * torch.randn(m, n, generator=g)).to(torch.bfloat16).to(adt).to(DEV)
                x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(adt).to(DEV)
                dW = arena[off:off + n * k].view(n, k)
                boff = off + up8(n * k)
                db = arena[boff:boff + n] if has_b else None
                ops.linear_bwd_w(dy, x, dW, alpha=alpha, db=db)
                ops_in.append((dy, x, off, boff, n, k, has_b, alpha))
                off = boff + up8(n)
            # (the [130 x 70] problem has rows of 70 floats: it misses the 16-byte staging conditions of the grouped kernel
            # and leaves at once, through the generic kernel - the others wait in the queue)
            if len(sel) == 1:
                assert torch.equal(arena, before), "a queued launch must not have run yet"
            ops.wgrad_group_end()
            ops._gemm_record = None
            torch.cuda.synchronize()
            ngroup = sum(1 for r in rec if isinstance(r[0], dict) and r[0]["kind"].startswith("group"))
            assert ngroup == (0 if len(sel) == 1 else (2 if prec == "fp32" else 1)), ngroup
            for dy, x, off, boff, n, k, has_b, alpha in ops_in:
                want = before[off:off + n * k].view(n, k).double().cpu() + alpha * dy.double().cpu().t() @ x.double().cpu()
                report("grouped dW [%d x %d], %d rows (%s)" % (n, k, dy.shape[0], prec), arena[off:off + n * k].view(n, k), want, 2e-5)
                if has_b:
                    wb = before[boff:boff + n].double().cpu() + alpha * dy.double().cpu().sum(0)
                    report("grouped db [%d] (%s)" % (n, prec), arena[boff:boff + n], wb, 2e-5)
                else:
                    assert torch.equal(arena[boff:boff + n], before[boff:boff + n])
    finally:
        ops._gemm_record = None
        ops.wgrad_group_end()
        espnet_amd.set_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_grouped_weight_gradients_same_destination(ops, prec):
    """several queued weight gradients that accumulate into the SAME dW - a weight used more than once
    per backward pass with 65..511 reduction rows (split-K 1 = plain read-modify-write tiles) and mixes of plain and
    atomic problems - must not share one grouped launch; the result equals the float64 sum of all contributions"""
    import espnet_amd
    espnet_amd.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(11)
        adt = ops.act_dtype()
        n, k = 320, 256
        arena = torch.randn(n * k + n + 64 * 64, generator=g).to(DEV)
        before = arena.clone()
        ops.register_grad_arena(arena)
        dW, db = arena[:n * k].view(n, k), arena[n * k:n * k + n]
        other = arena[n * k + n:].view(64, 64)
        ops.wgrad_group_begin()
        terms, oterms = [], []
        for m in (128, 128, 96, 2000, 128, 300, 128):      # 2000 rows: split-K > 1 (atomics) between the plain ones
            dy = (0.1 * torch.randn(m, n, generator=g)).to(torch.bfloat16).to(adt).to(DEV)
            x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(adt).to(DEV)
            ops.linear_bwd_w(dy, x, dW, db=db)
            terms.append((dy, x))
            dy2 = (0.1 * torch.randn(m, 64, generator=g)).to(torch.bfloat16).to(adt).to(DEV)     # a second weight in between
            x2 = torch.randn(m, 64, generator=g).to(torch.bfloat16).to(adt).to(DEV)
            ops.linear_bwd_w(dy2, x2, other)
            oterms.append((dy2, x2))
        ops.wgrad_group_end()
        torch.cuda.synchronize()
        want = before[:n * k].view(n, k).double().cpu() + sum(a.double().cpu().t() @ b.double().cpu() for a, b in terms)
        report("dW shared by 7 queued problems (%s)" % prec, dW, want, 2e-5)
        wb = before[n * k:n * k + n].double().cpu() + sum(a.double().cpu().sum(0) for a, _ in terms)
        report("db shared by 7 queued problems (%s)" % prec, db, wb, 2e-5)
        wo = before[n * k + n:].view(64, 64).double().cpu() + sum(a.double().cpu().t() @ b.double().cpu() for a, b in oterms)
        report("second dW (%s)" % prec, other, wo, 2e-5)
    finally:
        ops.wgrad_group_end()
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("shape", [(32, 249, 256, 31), (3, 70, 96, 15), (2, 33, 300, 7), (1, 5, 32, 3)])
def test_dwconv_glu_fused_vs_float64(ops, shape):
    """eamd_dwconv_glu_fwd / _bwd_x / _bwd_w (GLU formed on load, its derivative applied in the input-gradient store)
    against torch float64: glu -> depthwise conv1d forward, and the gradients w.r.t. the pointwise-conv output a, the
    depthwise weight and bias; fp32 and bf16 da"""
    B, T, Cc, K = shape
    g = torch.Generator().manual_seed(T + K)
    a = torch.randn(B, T, 2 * Cc, generator=g)
    w, bias, dy = 0.3 * torch.randn(Cc, 1, K, generator=g), torch.randn(Cc, generator=g), torch.randn(B, T, Cc, generator=g)
    ad, wd_, bd_ = a.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    gl = torch.nn.functional.glu(ad, dim=-1)
    yr = torch.nn.functional.conv1d(gl.transpose(1, 2), wd_, bd_, padding=(K - 1) // 2, groups=Cc).transpose(1, 2)
    yr.backward(dy.double())
    a2, wk = a.view(B * T, 2 * Cc).to(DEV), w.view(Cc, K).to(DEV)
    y = ops.dwconv_glu_fwd(a2, wk, bias.to(DEV), B, T, Cc, K)
    report("dwconv_glu_fwd %s" % (shape,), y.view(B, T, Cc), yr, 2e-6)
    # with the BatchNorm statistics of the output from the same launch (partials in the epilogue + eamd_bn_finalize)
    rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    y2, mean, rstd = ops.dwconv_glu_fwd(a2, wk, bias.to(DEV), B, T, Cc, K, bn=(1e-5, 0.1, rm, rv, nbt))
    assert torch.equal(y2, y) and int(nbt) == 1
    yd = yr.detach().reshape(-1, Cc)
    report("dwconv_glu_fwd bn mean %s" % (shape,), mean, yd.mean(0), 1e-5)
    report("dwconv_glu_fwd bn rstd %s" % (shape,), rstd, (yd.var(0, unbiased=False) + 1e-5).rsqrt(), 1e-5)
    report("dwconv_glu_fwd bn running_var %s" % (shape,), rv, 0.9 + 0.1 * yd.var(0, unbiased=True), 1e-5)
    dyd = dy.view(B * T, Cc).to(DEV)
    da = ops.dwconv_glu_bwd_x(dyd, wk, a2, B, T, Cc, K)
    report("dwconv_glu_bwd_x %s" % (shape,), da.view(B, T, 2 * Cc), ad.grad, 2e-6)
    da16 = ops.dwconv_glu_bwd_x(dyd, wk, a2, B, T, Cc, K, torch.bfloat16)
    report("dwconv_glu_bwd_x bf16 %s" % (shape,), da16.float().view(B, T, 2 * Cc), ad.grad, 4e-3)
    dw, db = torch.zeros(Cc, K, device=DEV), torch.zeros(Cc, device=DEV)
    ops.dwconv_glu_bwd_w(dyd, a2, dw, db, B, T, Cc, K)
    report("dwconv_glu_bwd_w %s" % (shape,), dw, wd_.grad.view(Cc, K), 1e-5)
    report("dwconv_glu_bwd_w bias %s" % (shape,), db, bd_.grad, 1e-5)


@pytest.mark.parametrize("shape", [(4, 80, 128, 31), (2, 33, 300, 29), (3, 17, 64, 3), (1, 5, 32, 1)])
def test_dwconv_kernel_sizes(ops, shape):
    """depthwise conv fwd / input grad / weight grad at the recipe's kernel size 31 and at ragged channel counts"""
    B, T, Cc, K = shape
    g = torch.Generator().manual_seed(K)
    x, w = torch.randn(B, T, Cc, generator=g), torch.randn(Cc, 1, K, generator=g)
    bias, dy = torch.randn(Cc, generator=g), torch.randn(B, T, Cc, generator=g)
    xd, wd_, bd_ = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    yr = torch.nn.functional.conv1d(xd.transpose(1, 2), wd_, bd_, padding=(K - 1) // 2, groups=Cc).transpose(1, 2)
    yr.backward(dy.double())
    wk = w.view(Cc, K).to(DEV)
    report("dwconv_fwd %s" % (shape,), ops.dwconv_fwd(x.to(DEV), wk, bias.to(DEV), B, T, Cc, K), yr, 1e-6)
    report("dwconv_bwd_x %s" % (shape,), ops.dwconv_bwd_x(dy.to(DEV), wk, B, T, Cc, K), xd.grad, 1e-6)
    dw, db = torch.zeros(Cc, K, device=DEV), torch.zeros(Cc, device=DEV)
    ops.dwconv_bwd_w(dy.to(DEV), x.to(DEV), dw, db, B, T, Cc, K)
    report("dwconv_bwd_w %s" % (shape,), dw, wd_.grad.view(Cc, K), 1e-5)
    report("dwconv_bwd_b %s" % (shape,), db, bd_.grad, 1e-5)


@pytest.mark.parametrize("shape", [(2100, 1100, 256), (4100, 1500, 256), (3000, 2200, 512), (7968, 2048, 256)])
def test_gemm_persistent_tiles(ops, shape):
    """launches with >= 512 tiles and K % 256 == 0 take the persistent kernel (gemm_persist.hip): all four
    operand layouts and the fused epilogues against a float64 product of the same bf16 operands"""
    import espnet_amd
    espnet_amd.set_precision("bf16")
    try:
        M, N, K = shape
        g = torch.Generator().manual_seed(M + N)
        bf = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16)
        A, At, B, Bt = bf(M, K), bf(K, M), bf(N, K), bf(K, N)
        bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        for ta, tb in ((0, 0), (0, 1), (1, 0), (1, 1)):
            a, b = (At if ta else A), (Bt if tb else B)
            want = (a.double().t() if ta else a.double()) @ (b.double() if tb else b.double().t())
            C = torch.empty(M, N, device=DEV)
            ops.gemm(a.to(DEV), b.to(DEV), C, M, N, K, M if ta else K, N if tb else K, N, transA=ta, transB=tb,
                     bias=bias.to(DEV), R=R.to(DEV), ldr=N, alpha=0.5)
            report("persistent gemm ta%d tb%d %s" % (ta, tb, shape), C, 0.5 * (want + bias.double()) + R.double(), 2e-6)
        # bf16 output + Swish-derivative mask of a bf16 aux operand + fused dropout (the FFN backward epilogue)
        aux = bf(M, N)
        Cb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm(A.to(DEV), Bt.to(DEV), Cb, M, N, K, K, N, N, transB=1, epilogue=4, aux=aux.to(DEV), ldaux=N,
                 drop=(0.2, 9))
        plain = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), Bt.to(DEV), plain, M, N, K, K, N, N, transB=1, epilogue=4, aux=aux.to(DEV), ldaux=N)
        want = ops.dropout(plain, 0.2, 9)
        report("persistent gemm dswish+dropout bf16 out", Cb.float(), want, 4e-3)
        # dual output
        z = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        h = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.linear_fwd(A.to(DEV), B.to(DEV), bias.to(DEV), out=z, drop=(0.1, 3), Hb=h, h_act=ops.ACT_SWISH)
        ref = ops.linear_fwd(A.to(DEV), B.to(DEV), bias.to(DEV))
        report("persistent gemm dual z", z.float(), ref, 4e-3)
        report("persistent gemm dual h", h.float(), ops.dropout(ref, 0.1, 3, act=ops.ACT_SWISH), 4e-3)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("dims", [(32, 249, 249, 4, 64), (3, 101, 249, 4, 64), (5, 7, 300, 2, 64), (9, 130, 65, 2, 128)])
def test_gemm_direct_short_k(ops, dims):
    """k-contiguous bf16 products with K = 64 / 128 and a plain fp32 result take the register-direct kernel
    (gemm_direct.hip): attention-score layout [B,T,H,dk] x [B,T,H,dk] -> [H,B,T1,ldp], ragged edges, batch counts
    that are not multiples of the 8 XCDs, alpha; pad columns of the result stay untouched"""
    Bb, T1, T2, H, dk = dims
    D = H * dk
    g = torch.Generator().manual_seed(T1 + T2)
    q, k = ((torch.randn(Bb, T, D, generator=g)).to(torch.bfloat16) for T in (T1, T2))
    ldp = (T2 + 7) // 8 * 8
    sc = torch.full((H * Bb * T1 * ldp,), -7.0, device=DEV)
    ops.gemm(q.to(DEV), k.to(DEV), sc, T1, T2, dk, D, D, ldp, batch=(Bb, H), sA=(T1 * D, dk), sB=(T2 * D, dk),
             sC=(T1 * ldp, Bb * T1 * ldp), alpha=0.25)
    ref = 0.25 * torch.einsum("bihd,bjhd->hbij", q.view(Bb, T1, H, dk).double(), k.view(Bb, T2, H, dk).double())
    got = sc.view(H, Bb, T1, ldp)
    report("gemm_direct %s" % (dims,), got[..., :T2], ref, 3e-6)
    assert bool((got[..., T2:] == -7.0).all())
    # unbatched, M and N below one wave tile
    a, b = (torch.randn(n, 64, generator=g).to(torch.bfloat16) for n in (5, 19))
    C = torch.empty(5, 19, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), C, 5, 19, 64, 64, 64, 19)
    report("gemm_direct tiny", C, a.double() @ b.double().t(), 3e-6)


@pytest.mark.parametrize("case", [
    dict(B=3, T1=249, T2=249, rel=True, mask="len"), dict(B=2, T1=256, T2=256, rel=True, mask=None),
    dict(B=9, T1=30, T2=30, rel=True, mask="len"), dict(B=2, T1=65, T2=65, rel=True, mask="len"),
    dict(B=2, T1=101, T2=249, rel=False, mask="len"), dict(B=3, T1=101, T2=101, rel=False, mask="causal"),
    dict(B=2, T1=1, T2=77, rel=False, mask="len"), dict(B=2, T1=64, T2=16, rel=False, mask="dead")])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_attention_forward_fused_matches_unfused(ops, case, prec):
    """eamd_attn_fwd (scores + rel-shift + mask + softmax + context in one launch) against the score GEMMs,
    eamd_softmax_fwd and the context GEMM on the same bf16 operands: probabilities and context agree to bf16 rounding,
    fully masked rows give zeros, pad columns of P are zero; q / k / v are column blocks of one fused [B*T, 3D] buffer
    where the layout allows it (self-attention), as in the model.  fp32 mode: eamd_attn_fwd_f32 against the same path in
    fp32 (agreement to fp32 rounding)"""
    import espnet_amd
    from espnet_amd import functional as F_
    espnet_amd.set_precision(prec)
    tol_p, tol_c, tol_abs = (3e-3, 6e-3, 2e-2) if prec == "bf16" else (3e-6, 3e-6, 2e-6)
    try:
        B, T1, T2, rel, mk = case["B"], case["T1"], case["T2"], case["rel"], case["mask"]
        H, dk = 4, 64
        D = H * dk
        g = torch.Generator().manual_seed(T1 * 7 + T2)
        bf = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(ops.act_dtype()).to(DEV)
        if T1 == T2:
            qkv = bf(B * T1, 3 * D)
            k, v = F_._MV(qkv, D, 3 * D), F_._MV(qkv, 2 * D, 3 * D)
            qu = bf(B * T1, D) if rel else F_._MV(qkv, 0, 3 * D)
        else:
            qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
        qv = bf(B * T1, D) if rel else None
        p = bf(T2, D) if rel else None
        mask = None
        if mk == "len":
            lens = torch.linspace(T2, max(1, T2 // 2), B).long()
            mask = (torch.arange(T2)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T2).contiguous().to(DEV)
        elif mk == "causal":
            mask = torch.tril(torch.ones(T1, T2)).to(torch.uint8).expand(B, T1, T2).contiguous().to(DEV)
        elif mk == "dead":
            mask = torch.ones(B, 1, T2, dtype=torch.uint8)
            mask[1] = 0
            mask = mask.to(DEV)
        fused = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk)
        assert fused is not None
        P1, _Pd, c1 = fused
        P0 = F_.attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
        c0 = F_.attn_context_fwd(P0, v, B, T1, T2, H, dk)
        ldp = F_._ldp(T2)
        P0v, P1v = P0.view(H, B, T1, ldp).float(), P1.view(H, B, T1, ldp).float()
        report("fused attention P %s %s" % (prec, case), P1v, P0v, tol_p)
        report("fused attention ctx %s %s" % (prec, case), c1.float(), c0.float(), tol_c)
        assert float((P1v - P0v).abs().max()) <= tol_abs
        assert bool((P1v[..., T2:] == 0).all())
        if mk == "dead":
            assert bool((P1v[:, 1] == 0).all()) and bool((c1.view(B, T1, D)[1] == 0).all())
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("case", [
    dict(B=3, T1=249, T2=249, rel=True, qkv=True), dict(B=2, T1=256, T2=256, rel=True, qkv=False),
    dict(B=9, T1=30, T2=30, rel=True, qkv=True), dict(B=2, T1=65, T2=65, rel=False, qkv=True),
    dict(B=2, T1=101, T2=249, rel=False, qkv=False), dict(B=3, T1=101, T2=101, rel=False, qkv=False),
    dict(B=2, T1=300, T2=300, rel=True, qkv=True), dict(B=2, T1=70, T2=333, rel=False, qkv=False)])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_attention_backward_fused_matches_unfused(ops, case, prec):
    """eamd_attn_bwd_q (score gradient + softmax backward + inverse rel-shift scatter + dq in one launch) inside
    attn_core_bwd against the GEMM / eamd_softmax_bwd / GEMM path: every returned gradient (dq, dqv, dk, dv, dpos)
    agrees to bf16 rounding, with q / k / v gradients written into the fused [B*T, 3D] buffer where the model does so.
    fp32 mode: eamd_attn_bwd_q_f32, agreement to fp32 rounding"""
    import espnet_amd
    from espnet_amd import functional as F_
    espnet_amd.set_precision(prec)
    tol = 6e-3 if prec == "bf16" else 1e-5
    try:
        B, T1, T2, rel, use_qkv = case["B"], case["T1"], case["T2"], case["rel"], case["qkv"]
        H, dk = 4, 64
        D = H * dk
        g = torch.Generator().manual_seed(T1 * 5 + T2)
        bf = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(ops.act_dtype()).to(DEV)
        if use_qkv:
            qkv = bf(B * T1, 3 * D)
            k, v = F_._MV(qkv, D, 3 * D), F_._MV(qkv, 2 * D, 3 * D)
            qu = bf(B * T1, D) if rel else F_._MV(qkv, 0, 3 * D)
        else:
            qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
        qv = bf(B * T1, D) if rel else None
        p = bf(T2, D) if rel else None
        lens = torch.linspace(T2, max(1, T2 // 2), B).long()
        mask = (torch.arange(T2)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T2).contiguous().to(DEV)
        P = F_.attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
        dctx = bf(B * T1, D)
        outs = {}
        for fuse in (True, False):
            F_.FUSE_ATTN = fuse
            dqkv = torch.zeros(B * T1, 3 * D, device=DEV, dtype=ops.act_dtype()) if use_qkv else None
            r = F_.attn_core_bwd(dctx, P.clone(), qu, qv, k, v, p, B, T1, T2, H, dk, dqkv=dqkv)
            outs[fuse] = [None if x is None else x.float().clone() for x in r] + [None if dqkv is None else dqkv.float().clone()]
        names = ("dqu", "dqv", "dk", "dv", "dpos", "dqkv")
        for n, a, b in zip(names, outs[True], outs[False]):
            assert (a is None) == (b is None), n
            if a is not None:
                report("fused attention bwd %s %s %s" % (n, prec, case), a, b, tol)
    finally:
        F_.FUSE_ATTN = True
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("case", [dict(B=3, T1=249, T2=249, rel=True), dict(B=2, T1=101, T2=249, rel=False),
                                  dict(B=2, T1=65, T2=65, rel=True), dict(B=3, T1=101, T2=101, rel=False)])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_attention_dropout_fused_matches_unfused(ops, case, prec):
    """attention dropout (attention.py:91) INSIDE the fused kernels: the forward builds the context from
    Pd = dropout(P) with the very mask eamd_dropout draws on P (same keep pattern as ops.dropout(P)), keeps P undropped
    for the softmax backward, and eamd_attn_bwd_q masks dP the same way - against the GEMM / softmax / dropout / GEMM
    path: context and every gradient agree to the mode's rounding"""
    import espnet_amd
    from espnet_amd import functional as F_
    espnet_amd.set_precision(prec)
    tol_f, tol_g = (6e-3, 8e-3) if prec == "bf16" else (3e-6, 1e-5)
    try:
        B, T1, T2, rel = case["B"], case["T1"], case["T2"], case["rel"]
        H, dk = 4, 64
        D = H * dk
        g = torch.Generator().manual_seed(T1 * 13 + T2)
        bf = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(torch.bfloat16).to(ops.act_dtype()).to(DEV)
        qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
        qv = bf(B * T1, D) if rel else None
        p = bf(T2, D) if rel else None
        lens = torch.linspace(T2, max(1, T2 // 2), B).long()
        mask = (torch.arange(T2)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T2).contiguous().to(DEV)
        drop = (0.2, 90210)
        ops.manual_seed(5)
        P1, Pd1, c1 = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk, drop=drop)
        P0 = F_.attn_scores_fwd(qu, qv, k, p, mask, B, T1, T2, H, dk)
        Pd0 = ops.dropout(P0, *drop)
        c0 = F_.attn_context_fwd(Pd0, v, B, T1, T2, H, dk)
        ldp = F_._ldp(T2)
        view = lambda t: t.view(H, B, T1, ldp)[..., :T2].float()
        report("attention dropout: P %s %s" % (prec, case), view(P1), view(P0), tol_f)
        nz = view(P0) > 1e-4            # where the probability itself is not (near) zero the keep pattern must be identical
        assert torch.equal((view(Pd1) != 0)[nz], (view(Pd0) != 0)[nz])
        kept = float((view(Pd1) != 0)[nz].float().mean())
        assert abs(kept - 0.8) < 0.02, kept
        report("attention dropout: Pd %s %s" % (prec, case), view(Pd1), view(Pd0), tol_f)
        report("attention dropout: ctx %s %s" % (prec, case), c1.float(), c0.float(), tol_f)
        dctx = bf(B * T1, D)
        outs = {}
        for fuse in (True, False):
            F_.FUSE_ATTN = fuse
            r = F_.attn_core_bwd(dctx, P0.clone(), qu, qv, k, v, p, B, T1, T2, H, dk, Pd=Pd0, attn_drop=drop)
            outs[fuse] = [None if x is None else x.float().clone() for x in r]
        for n, a, b in zip(("dqu", "dqv", "dk", "dv", "dpos"), outs[True], outs[False]):
            assert (a is None) == (b is None), n
            if a is not None:
                report("attention dropout bwd %s %s %s" % (n, prec, case), a, b, tol_g)
    finally:
        F_.FUSE_ATTN = True
        espnet_amd.set_precision("fp32")


def _oracle_attention(oracle, qu, qv, k, v, p, mask, dctx, B, T1, T2, H, dk):
    """float64 restatement of the reference's attention core on the SAME bf16-rounded operands, built from the
    oracle's pieces: scores (+ oracle.rel_shift of the position term, attention.py:141-162,195-203), the mask ->
    finfo.min -> softmax -> zero-fill sequence of forward_attention (attention.py:63-92) and its autograd gradients."""
    f64 = lambda t, T: t.detach().cpu().double().view(-1, T, H, dk).requires_grad_(True)
    Q, K, V = f64(qu, T1), f64(k, T2), f64(v, T2)
    QV = f64(qv, T1) if qv is not None else None
    Pm = p.detach().cpu().double().view(1, T2, H, dk).requires_grad_(True) if p is not None else None
    sc = torch.einsum("bihd,bjhd->bhij", Q, K)
    if p is not None:
        sc = sc + oracle.rel_shift(torch.einsum("bihd,xjhd->bhij", QV, Pm))
    sc = sc / math.sqrt(dk)
    if mask is not None:
        m = mask.cpu().bool().unsqueeze(1).eq(0)                     # (B, 1, 1|T1, T2)
        attn = torch.softmax(sc.masked_fill(m, torch.finfo(torch.float32).min), dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(sc, dim=-1)
    ctx = torch.einsum("bhij,bjhd->bihd", attn, V).reshape(B * T1, H * dk)
    ctx.backward(dctx.detach().cpu().double())
    g = lambda t: None if t is None else t.grad
    return attn.detach(), ctx.detach(), g(Q), g(QV), g(K), g(V), g(Pm)


@pytest.mark.parametrize("case", [
    dict(B=3, T1=249, T2=249, rel=True, mask="len"), dict(B=2, T1=101, T2=101, rel=False, mask="causal"),
    dict(B=2, T1=101, T2=249, rel=False, mask="len"), dict(B=2, T1=1, T2=77, rel=False, mask="len"),
    dict(B=3, T1=64, T2=64, rel=True, mask="dead"), dict(B=2, T1=30, T2=30, rel=True, mask=None),
    dict(B=2, T1=64, T2=16, rel=False, mask="dead"),
    dict(B=2, T1=374, T2=374, rel=True, mask="len"), dict(B=1, T1=512, T2=512, rel=True, mask=None),     # 32 key tiles (config 5: T' = 374)
    dict(B=2, T1=101, T2=500, rel=False, mask="len")])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_attention_fused_vs_oracle(ops, oracle, case, prec):
    """The kernels bench.py dispatches at config 2 (eamd_attn_fwd / eamd_attn_bwd_q: H = 4, d_k = 64, T2 <= 256)
    against the ORACLE in float64 on the same bf16 operands - not against our own unfused path: probabilities,
    context and every gradient (dq+u, dq+v, dk, dv, dpos), ragged key lengths, a causal mask, a single query and a
    fully masked utterance (reference: zeros, attention.py:84-88).  Error budget: P and dS are rounded to bf16
    inside the kernels (2^-9 relative), so rel-L2 <= 4e-3 on P / ctx and <= 8e-3 on the gradients.
    fp32 mode (the bench headline): eamd_attn_fwd_f32 / eamd_attn_bwd_q_f32 on the same operand values held in fp32;
    everything stays fp32 (fp32 MFMA, v_exp_f32 at 1 ulp), so rel-L2 <= 2e-6 on P / ctx and <= 1e-5 on the gradients."""
    import espnet_amd
    from espnet_amd import functional as F_
    espnet_amd.set_precision(prec)
    tol_f, tol_g = (4e-3, 8e-3) if prec == "bf16" else (2e-6, 1e-5)
    try:
        B, T1, T2, rel, mk = case["B"], case["T1"], case["T2"], case["rel"], case["mask"]
        H, dk = 4, 64
        D = H * dk
        g = torch.Generator().manual_seed(T1 * 11 + T2)
        bf = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(torch.bfloat16).to(ops.act_dtype()).to(DEV)
        qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
        qv = bf(B * T1, D) if rel else None
        p = bf(T2, D) if rel else None
        mask = None
        if mk == "len":
            lens = torch.linspace(T2, max(1, T2 // 2), B).long()
            mask = (torch.arange(T2)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T2).contiguous().to(DEV)
        elif mk == "causal":
            mask = torch.tril(torch.ones(T1, T2)).to(torch.uint8).expand(B, T1, T2).contiguous().to(DEV)
        elif mk == "dead":
            mask = torch.ones(B, 1, T2, dtype=torch.uint8)
            mask[1] = 0
            mask[0, 0, T2 // 2:] = 0
            mask = mask.to(DEV)
        assert ops.attn_fwd_supported(T1, T2, dk, rel)
        fused = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk)
        assert fused is not None, "eamd_attn_fwd declined the config-2 operand layout"
        P1, _Pd, c1 = fused
        assert P1.dtype == c1.dtype == ops.act_dtype()
        dctx = bf(B * T1, D)
        F_.FUSE_ATTN = True
        dqu, dqv, dkk, dvv, dpos = F_.attn_core_bwd(dctx, P1, qu, qv, k, v, p, B, T1, T2, H, dk)
        attn, ctx, gq, gqv, gk, gv, gp = _oracle_attention(oracle, qu, qv, k, v, p, mask, dctx, B, T1, T2, H, dk)
        ldp = F_._ldp(T2)
        Pv = P1.view(H, B, T1, ldp).float().permute(1, 0, 2, 3)
        report("attn_fwd P vs oracle %s %s" % (prec, case), Pv[..., :T2], attn, tol_f)
        assert bool((Pv[..., T2:] == 0).all())
        report("attn_fwd ctx vs oracle %s %s" % (prec, case), c1.float(), ctx, tol_f)
        if mk == "dead":
            assert bool((Pv[1] == 0).all()) and bool((c1.view(B, T1, D)[1] == 0).all())
            assert bool((Pv[0, ..., T2 // 2:] == 0).all())
        r2 = lambda t, T: t.reshape(B * T, D)
        report("attn_bwd dq(u) vs oracle %s %s" % (prec, case), dqu.float(), r2(gq, T1), tol_g)
        report("attn_bwd dk vs oracle %s %s" % (prec, case), dkk.float(), r2(gk, T2), tol_g)
        report("attn_bwd dv vs oracle %s %s" % (prec, case), dvv.float(), r2(gv, T2), tol_g)
        if rel:
            report("attn_bwd dq(v) vs oracle %s %s" % (prec, case), dqv.float(), r2(gqv, T1), tol_g)
            report("attn_bwd dpos vs oracle %s %s" % (prec, case), dpos.float().view(T2, D), gp.reshape(T2, D), tol_g)
    finally:
        F_.FUSE_ATTN = True
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("case", [
    dict(B=2, T1=999, T2=999, rel=True, mask="len"), dict(B=1, T1=1500, T2=1500, rel=True, mask=None),
    dict(B=2, T1=101, T2=999, rel=False, mask="len"), dict(B=2, T1=600, T2=600, rel=True, mask="dead"),
    dict(B=1, T1=530, T2=530, rel=False, mask="causal"), dict(B=1, T1=2048, T2=2048, rel=True, mask="len"),
    dict(B=1, T1=2500, T2=2500, rel=True, mask="len"), dict(B=2, T1=101, T2=4096, rel=False, mask="len"),
    dict(B=1, T1=2100, T2=2100, rel=False, mask="causal"), dict(B=1, T1=4096, T2=4096, rel=True, mask=None),
    dict(B=2, T1=1030, T2=1030, rel=True, mask="len"), dict(B=1, T1=1001, T2=1001, rel=False, mask="causal"),
    dict(B=2, T1=77, T2=1090, rel=False, mask="dead")])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_attention_fused_long_rows_vs_oracle(ops, oracle, case, prec):
    """Rows of 513 .. 4096 keys stay on the fused path in both precision modes (attn_fwd_long_kernel / attn_bwd_q_long_kernel:
    16 queries per workgroup up to 2048 keys - 32 between 977 and 1100 keys, two query tiles per wave -, 8 beyond; every phase split
    over the keys; bf16 operands are widened on load):
    probabilities, context and every gradient against the oracle in float64, the legacy rel_shift at T1 = T2 = 999 / 1500 / 2048 /
    2500 / 4096, cross-attention 101 x 999 and 101 x 4096, causal masks and a fully masked utterance.  Tolerances as test_attention_fused_vs_oracle.
    reference: transformer/attention.py:63-92, 141-206."""
    import espnet_amd
    from espnet_amd import functional as F_
    espnet_amd.set_precision(prec)
    try:
        _long_attention_case(ops, oracle, case, prec, F_)
    finally:
        espnet_amd.set_precision("fp32")


def _long_attention_case(ops, oracle, case, prec, F_):
    tol_f, tol_g = (4e-3, 8e-3) if prec == "bf16" else (2e-6, 1e-5)
    B, T1, T2, rel, mk = case["B"], case["T1"], case["T2"], case["rel"], case["mask"]
    H, dk = 4, 64
    D = H * dk
    g = torch.Generator().manual_seed(T1 * 11 + T2)
    bf = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(torch.bfloat16).to(ops.act_dtype()).to(DEV)
    qu, k, v = bf(B * T1, D), bf(B * T2, D), bf(B * T2, D)
    qv = bf(B * T1, D) if rel else None
    p = bf(T2, D) if rel else None
    mask = None
    if mk == "len":
        lens = torch.linspace(T2, max(1, T2 // 2), B).long()
        mask = (torch.arange(T2)[None, :] < lens[:, None]).to(torch.uint8).view(B, 1, T2).contiguous().to(DEV)
    elif mk == "causal":
        mask = torch.tril(torch.ones(T1, T2)).to(torch.uint8).expand(B, T1, T2).contiguous().to(DEV)
    elif mk == "dead":
        mask = torch.ones(B, 1, T2, dtype=torch.uint8)
        mask[1] = 0
        mask[0, 0, T2 // 2:] = 0
        mask = mask.to(DEV)
    assert ops.attn_fwd_supported(T1, T2, dk, rel)
    fused = F_.attn_fwd_fused(qu, qv, k, v, p, mask, B, T1, T2, H, dk)
    assert fused is not None, "eamd_attn_fwd_f32 declined a long row"
    P1, _Pd, c1 = fused
    dctx = bf(B * T1, D)
    took, orig = [], ops.attn_bwd_q

    def spy(*a_, **k_):
        took.append(orig(*a_, **k_))
        return took[-1]
    ops.attn_bwd_q = spy
    try:
        dqu, dqv, dkk, dvv, dpos = F_.attn_core_bwd(dctx, P1, qu, qv, k, v, p, B, T1, T2, H, dk)
    finally:
        ops.attn_bwd_q = orig
    assert took == [True], "the query-side backward left the fused path"
    attn, ctx, gq, gqv, gk, gv, gp = _oracle_attention(oracle, qu, qv, k, v, p, mask, dctx, B, T1, T2, H, dk)
    ldp = F_._ldp(T2)
    Pv = P1.view(H, B, T1, ldp).float().permute(1, 0, 2, 3)
    report("long attn P vs oracle %s" % case, Pv[..., :T2], attn, tol_f)
    assert bool((Pv[..., T2:] == 0).all())
    report("long attn ctx vs oracle %s" % case, c1.float(), ctx, tol_f)
    if mk == "dead":
        assert bool((Pv[1] == 0).all()) and bool((c1.view(B, T1, D)[1] == 0).all())
    r2 = lambda t, T: t.reshape(B * T, D)
    report("long attn dq(u) vs oracle %s" % case, dqu.float(), r2(gq, T1), tol_g)
    report("long attn dk vs oracle %s" % case, dkk.float(), r2(gk, T2), tol_g)
    report("long attn dv vs oracle %s" % case, dvv.float(), r2(gv, T2), tol_g)
    if rel:
        report("long attn dq(v) vs oracle %s" % case, dqv.float(), r2(gqv, T1), tol_g)
        report("long attn dpos vs oracle %s" % case, dpos.float().view(T2, D), gp.reshape(T2, D), tol_g)


@pytest.mark.parametrize("shape", [(7968, 256, 768), (333, 64, 64), (70, 320, 320), (5, 512, 1024)])
def test_add_cast_colsum2(ops, shape):
    """one pass: out = bf16(a + b) into a column block, both column sums accumulated"""
    rows, D, ld = shape
    g = torch.Generator().manual_seed(rows)
    a, b = torch.randn(rows, D, generator=g).to(DEV), torch.randn(rows, D, generator=g).to(DEV)
    sa, sb = torch.ones(D, device=DEV), torch.full((D,), -2.0, device=DEV)
    out = torch.zeros(rows, ld, device=DEV, dtype=torch.bfloat16)
    off = ld - D
    ops.add_cast_colsum2(a, b, sa, sb, out=out, out_off=off, ld_out=ld)
    want = (a + b).to(torch.bfloat16)
    assert torch.equal(out[:, off:], want)
    assert off == 0 or bool((out[:, :off] == 0).all())
    report("add_cast_colsum2 suma", sa, 1 + a.double().sum(0), 1e-5)
    report("add_cast_colsum2 sumb", sb, -2 + b.double().sum(0), 1e-5)
    # fp32 twin (eamd_add_colsum2_f32): exact sum, same column sums
    sa, sb = torch.ones(D, device=DEV), torch.full((D,), -2.0, device=DEV)
    out = torch.zeros(rows, ld, device=DEV)
    ops.add_cast_colsum2(a, b, sa, sb, out=out, out_off=off, ld_out=ld)
    assert torch.equal(out[:, off:], a + b)
    assert off == 0 or bool((out[:, :off] == 0).all())
    report("add_colsum2_f32 suma", sa, 1 + a.double().sum(0), 1e-5)
    report("add_colsum2_f32 sumb", sb, -2 + b.double().sum(0), 1e-5)


@pytest.mark.parametrize("prec,tile,M,N,K", [("fp32", 64, 150, 333, 64), ("fp32", 128, 300, 5000, 320), ("bf16", 64, 700, 5000, 320),
                                             ("bf16", 128, 2500, 5000, 320), ("bf16", 64, 4200, 1000, 256)])
def test_gemm_row_statistics_epilogue(ops, prec, tile, M, N, K):
    """epilogue 7 (eamd_gemm_t.stats): per-row, per-column-tile (max, sum exp) of v = A W^T + bias and two gathered columns,
    without v being stored; combined they give the row log-sum-exp.  Against torch on the same operands (float64)."""
    import espnet_amd
    espnet_amd.set_precision(prec)
    try:
        dt = ops.act_dtype()
        g = torch.Generator().manual_seed(M + N)
        a = (torch.randn(M, K, generator=g) * 0.7).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.3).to(dt).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        col = torch.randint(-1, N, (M,), generator=g).to(torch.int32).to(DEV)
        fix = 0
        tn = (N + tile - 1) // tile
        part = torch.full((M * tn * 2,), float("nan"), device=DEV)
        zcol = torch.full((M,), float("nan"), device=DEV)
        zfix = torch.full((M,), float("nan"), device=DEV)
        ops.gemm(a, w, None, M, N, K, K, K, N, bias=bias, epilogue=ops.EPI_ROW_STATS, tile=tile, stats=(part, col, zcol, zfix, fix))
        v = a.double() @ w.double().t() + bias.double()
        pm = part.view(M, tn, 2).double()
        mx = pm[:, :, 0].max(1).values
        lse = mx + torch.log((pm[:, :, 1] * torch.exp(pm[:, :, 0] - mx[:, None])).sum(1))
        report("row-stats lse %s tile %d" % (prec, tile), lse, torch.logsumexp(v, 1), 2e-6)
        report("row-stats fixed column", zfix, v[:, fix], 1e-5)
        has = col >= 0
        report("row-stats gathered column", zcol[has], v[has, col[has].long()], 1e-5)
        # every tile's maximum is the true maximum of its columns
        vt = torch.nn.functional.pad(v, (0, tn * tile - N), value=float("-inf")).view(M, tn, tile).max(2).values
        report("row-stats tile maxima", pm[:, :, 0], vt, 1e-5)
    finally:
        espnet_amd.set_precision("fp32")


# ---------------------------------------------------------------------------------------------
# fused position-wise feed-forward (eamd_ffn_fwd / eamd_ffn_bwd)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,F,act,p_in,p_out", [(7968, 2048, 2, 0.1, 0.1), (4100, 1024, 1, 0.0, 0.0), (4097, 256, 2, 0.1, 0.0),
                                                (33, 384, 2, 0.0, 0.2),
                                                (3232, 2048, 1, 0.1, 0.1), (1601, 512, 2, 0.1, 0.0)])      # hidden split in two
def test_ffn_fused_vs_float64(ops, M, F, act, p_in, p_out):
    """positionwise_feed_forward.py:12-32 + the block wiring of conformer/encoder_layer.py:96-103 as ONE launch, against
    float64: out = R + alpha * drop(drop(act(x W1^T + b1)) W2^T + b2), the two tensors kept for backward (h and the
    factor f = mask / (1 - p) * act'(z)), and the backward launch dz = alpha (dy W2) (.) f, dx = dz W1.  The dropout
    masks must be eamd_dropout's own (same salts, same step): they are drawn here by dropping a tensor of ones."""
    import espnet_amd
    espnet_amd.set_precision("fp32")
    g = torch.Generator().manual_seed(M + F)
    D = 256
    x = torch.randn(M, D, generator=g)
    w1 = torch.randn(F, D, generator=g) * 0.06
    b1 = torch.randn(F, generator=g) * 0.1
    w2 = torch.randn(D, F, generator=g) * 0.03
    b2 = torch.randn(D, generator=g) * 0.1
    R = torch.randn(M, D, generator=g)
    dy = torch.randn(M, D, generator=g)
    alpha = 0.5
    ops.manual_seed(5)
    s_in, s_out = 1234, 987
    dev = [t.to(DEV) for t in (x, w1, b1, w2, b2, R, dy)]
    xd, w1d, b1d, w2d, b2d, Rd, dyd = dev
    out, f, h = ops.ffn_fwd(xd, w1d, b1d, w2d, b2d, act=act, alpha=alpha, R=Rd, drop=(p_in, s_in, p_out, s_out))
    dz, dx = ops.ffn_bwd(dyd, w1d, w2d, f, alpha=alpha)
    out_nosave, f0, h0 = ops.ffn_fwd(xd, w1d, b1d, w2d, b2d, act=act, alpha=alpha, R=Rd, drop=(p_in, s_in, p_out, s_out), save=False)
    assert f0 is None and h0 is None and torch.equal(out_nosave, out)
    m_in = ops.dropout(torch.ones(M, F, device=DEV), p_in, s_in).double().cpu() if p_in > 0 else torch.ones(M, F, dtype=torch.float64)
    m_out = ops.dropout(torch.ones(M, D, device=DEV), p_out, s_out).double().cpu() if p_out > 0 else torch.ones(M, D, dtype=torch.float64)
    if p_in > 0:
        assert 0.85 < float((m_in > 0).double().mean()) < 0.95
    z = x.double() @ w1.double().t() + b1.double()
    if act == 2:
        sg = torch.sigmoid(z)
        a, d = z * sg, sg * (1 + z * (1 - sg))
    else:
        a, d = z.clamp_min(0), (z > 0).double()
    h_ref, f_ref = a * m_in, d * m_in
    out_ref = R.double() + alpha * ((h_ref @ w2.double().t() + b2.double()) * m_out)
    tag = "M=%d F=%d act=%d p=(%g,%g)" % (M, F, act, p_in, p_out)
    if act == 1:       # ReLU: pre-activations within fp32 rounding of zero may take either side
        sure = (z.abs() > 1e-4)
        assert torch.equal((h.double().cpu() != 0) & sure, (h_ref != 0) & sure)
        h_cmp, f_cmp = torch.where(sure, h.double().cpu(), h_ref), torch.where(sure, f.double().cpu(), f_ref)
    else:
        h_cmp, f_cmp = h, f
    report("ffn fused h  " + tag, h_cmp, h_ref, 2e-6)
    report("ffn fused f  " + tag, f_cmp, f_ref, 2e-6)
    report("ffn fused out " + tag, out, out_ref, 2e-6)
    dz_ref = alpha * (dy.double() @ w2.double()) * f.double().cpu()
    report("ffn fused dz " + tag, dz, dz_ref, 2e-6)
    report("ffn fused dx " + tag, dx, dz_ref @ w1.double(), 2e-6)


@pytest.mark.parametrize("M,F,act,p_in,p_out", [(7968, 2048, 2, 0.1, 0.1), (4100, 1024, 1, 0.0, 0.0), (33, 256, 2, 0.1, 0.2)])
def test_ffn_fused_bf16_vs_float64(ops, M, F, act, p_in, p_out):
    """the bf16-operand twin (eamd_ffn_fwd / _bwd with dtype 1, csrc/ffn_bf16.hip) against float64 on the SAME bf16-rounded
    operands: h and f (stored bf16: rounding 2^-9), out (fp32 accumulation of bf16 products, h rounded to bf16 in between),
    and the backward launch on the transposed weight copies - dz (bf16), dx (fp32)"""
    import espnet_amd
    espnet_amd.set_precision("bf16")
    try:
        g = torch.Generator().manual_seed(M + F + 1)
        D = 256
        rb = lambda t_: t_.to(torch.bfloat16).float()  # noqa: E731
        x, w1, w2 = rb(torch.randn(M, D, generator=g)), rb(torch.randn(F, D, generator=g) * 0.06), rb(torch.randn(D, F, generator=g) * 0.03)
        b1, b2 = torch.randn(F, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
        R, dy = torch.randn(M, D, generator=g), rb(torch.randn(M, D, generator=g))
        alpha = 0.5
        ops.manual_seed(6)
        s_in, s_out = 4321, 789
        bf = lambda t_: t_.to(DEV).to(torch.bfloat16)  # noqa: E731
        xd, w1d, w2d, dyd = bf(x), bf(w1), bf(w2), bf(dy)
        out, f, h = ops.ffn_fwd(xd, w1d, b1.to(DEV), w2d, b2.to(DEV), act=act, alpha=alpha, R=R.to(DEV), drop=(p_in, s_in, p_out, s_out))
        assert f.dtype == torch.bfloat16 and h.dtype == torch.bfloat16 and out.dtype == torch.float32
        packs = ops.ffn_pack(w1d, w2d)
        dz, dx = ops.ffn_bwd(dyd, w1d, w2d, f, alpha=alpha, packed=packs[2:])
        m_in = ops.dropout(torch.ones(M, F, device=DEV), p_in, s_in).double().cpu() if p_in > 0 else torch.ones(M, F, dtype=torch.float64)
        m_out = ops.dropout(torch.ones(M, D, device=DEV), p_out, s_out).double().cpu() if p_out > 0 else torch.ones(M, D, dtype=torch.float64)
        z = x.double() @ w1.double().t() + b1.double()
        if act == 2:
            sg = torch.sigmoid(z)
            a, d = z * sg, sg * (1 + z * (1 - sg))
        else:
            a, d = z.clamp_min(0), (z > 0).double()
        h_ref, f_ref = a * m_in, d * m_in
        tag = "bf16 M=%d F=%d act=%d p=(%g,%g)" % (M, F, act, p_in, p_out)
        if act == 1:
            sure = z.abs() > 1e-3
            h_cmp, f_cmp = torch.where(sure, h.double().cpu(), h_ref), torch.where(sure, f.double().cpu(), f_ref)
        else:
            h_cmp, f_cmp = h, f
        report("ffn fused h  " + tag, h_cmp, h_ref, 4e-3)
        report("ffn fused f  " + tag, f_cmp, f_ref, 4e-3)
        hq = h.double().cpu()                     # the second product reads the stored (rounded) h
        out_ref = R.double() + alpha * ((hq @ w2.double().t() + b2.double()) * m_out)
        report("ffn fused out " + tag, out, out_ref, 2e-5)
        fq_ = f.double().cpu()
        dz_ref = alpha * (dy.double() @ w2.double()) * fq_
        report("ffn fused dz " + tag, dz, dz_ref, 4e-3)
        report("ffn fused dx " + tag, dx, dz.double().cpu() @ w1.double(), 2e-5)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("shape", [(8, 199, 39, 256), (2, 29, 19, 64), (3, 257, 39, 128)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_conv2_input_gradient_classes_in_one_launch(ops, shape, prec):
    """eamd_gemm_multi: the four stride-parity products of Conv2dSubsampling's second convolution issued together are, bit for
    bit, the four eamd_gemm launches (first shape: the one-launch fp32 kernel; the others and bf16 operands: the library's own
    one-by-one route), and equal conv2d's input gradient in float64.  reference: transformer/subsampling.py:28-35"""
    import espnet_amd
    from espnet_amd import functional as Fn
    B, Hi, Wi, Cc = shape
    espnet_amd.set_precision(prec)
    try:
        adt = torch.bfloat16 if prec == "bf16" else torch.float32
        g = torch.Generator().manual_seed(5)
        Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
        w = (torch.randn(Cc, Cc, 3, 3, generator=g) / (3.0 * Cc ** 0.5)).to(DEV)
        y_in = torch.relu(torch.randn(B * Hi * Wi, Cc, generator=g)).to(DEV).to(adt)
        dy = torch.randn(B * Ho * Wo, Cc, generator=g).to(DEV).to(adt)
        _wf, wd = ops.conv2_weight_prep(w, adt)

        def run(per_call):
            keep, ops.GEMM_MULTI_MAX = ops.GEMM_MULTI_MAX, per_call
            try:
                dw, db = torch.zeros(Cc, Cc, 3, 3, device=DEV), torch.zeros(Cc, device=DEV)
                return Fn._conv3s2_bwd(dy, y_in, wd, dw, db, B, Hi, Wi, Ho, Wo, Cc, adt), dw
            finally:
                ops.GEMM_MULTI_MAX = keep
        (d4, dw4), (d1, dw1) = run(4), run(1)
        (d3, _dw3) = run(3)                                  # 3 + 1: a launch of three problems, then a single one
        torch.cuda.synchronize()
        assert torch.equal(d4, d1) and torch.equal(d3, d1)
        ref = torch.nn.grad.conv2d_input((B, Cc, Hi, Wi), w.double().cpu(),
                                         dy.double().cpu().view(B, Ho, Wo, Cc).permute(0, 3, 1, 2), stride=2)
        ref = ref.permute(0, 2, 3, 1).reshape(B * Hi * Wi, Cc) * (y_in.double().cpu() > 0)
        report("conv2 dX classes %s %s" % (shape, prec), d4, ref, 2e-5 if prec == "fp32" else 1e-2)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("M,F,act", [(7968, 2048, 2), (1000, 512, 1), (3232, 2048, 1)])
def test_ffn_layernorm_in_front(ops, prec, M, F, act):
    """eamd_ffn_t.ln_x: the feed-forward kernel normalises its own input rows.  Against eamd_layernorm_fwd followed by the
    kernel without it: row statistics and normalised rows (which backward reads) to rounding, block output likewise; and the
    normalised rows against float64.  reference: conformer/encoder_layer.py:96-103 (x + s * dropout(ff(norm_ff(x))))"""
    import espnet_amd
    espnet_amd.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(11)
        D = 256
        adt = torch.bfloat16 if prec == "bf16" else torch.float32
        x = (torch.randn(M, D, generator=g) * 3.0 + 0.5).to(DEV)
        gam, bet = (1.0 + 0.1 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
        w1 = (torch.randn(F, D, generator=g) / 16.0).to(DEV).to(adt)
        w2 = (torch.randn(D, F, generator=g) / (F ** 0.5)).to(DEV).to(adt)
        b1, b2 = (0.1 * torch.randn(F, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
        eps = 1e-12
        drop = (0.1, 77, 0.1, 78)
        y_ref, mean_ref, rstd_ref = ops.layernorm_fwd(x, gam, bet, eps, adt)
        out_ref, f_ref, h_ref = ops.ffn_fwd(y_ref, w1, b1, w2, b2, act=act, alpha=0.5, R=x, drop=drop)
        xn = torch.full((M, D), float("nan"), device=DEV).to(adt)
        mean, rstd = torch.full((M,), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
        out, f, h = ops.ffn_fwd(xn, w1, b1, w2, b2, act=act, alpha=0.5, R=x, drop=drop, ln=(x, gam, bet, eps, mean, rstd))
        torch.cuda.synchronize()
        tag = "%s M=%d F=%d act=%d" % (prec, M, F, act)
        report("ffn+ln mean " + tag, mean, mean_ref, 1e-6)
        report("ffn+ln rstd " + tag, rstd, rstd_ref, 1e-6)
        xd = x.double().cpu()
        y64 = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + eps) * gam.double().cpu() + bet.double().cpu()
        report("ffn+ln rows vs float64 " + tag, xn, y64, 1e-6 if prec == "fp32" else 4e-3)
        report("ffn+ln rows " + tag, xn, y_ref, 1e-6 if prec == "fp32" else 1e-3)
        report("ffn+ln out  " + tag, out, out_ref, 1e-5 if prec == "fp32" else 2e-3)
        assert torch.isfinite(out).all() and torch.isfinite(xn.float()).all()
        if prec == "fp32":            # same masks, same arithmetic on rows that differ in the last bit
            report("ffn+ln h    " + tag, h, h_ref, 1e-5)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("rows,n,k", [(10, 5000, 15), (1, 50000, 10), (320, 5000, 10), (32, 50000, 10), (3, 7, 7), (5, 6144, 64), (4, 6145, 3),
                                      (6, 100, 10), (3, 300, 64), (2, 1, 1)])
def test_topk_rows(ops, rows, n, k):
    """eamd_topk_rows against torch.topk (the selections of a beam step: beam_search.py:143-176): values equal element for
    element; indices equal where the values are distinct; ties (planted duplicates, -inf runs of dead beam slots, NaN = -inf) in
    ascending index order; two launches bit-equal."""
    g = torch.Generator().manual_seed(rows * 131 + n)
    x = torch.randn(rows, n, generator=g)
    if n >= 100:
        x[:, 17] = x[:, 3]                               # a duplicate value
        x[0, : n // 2] = -float("inf")                   # dead slots
        x[-1, 5] = float("nan")
    if rows > 2:
        x[1] = -float("inf")                             # a row of dead slots only: the first k indices
    xd = x.to(DEV)
    v, i = ops.topk_rows(xd, k)
    v2, i2, j2 = ops.topk_rows(xd, k, idx32=True)
    torch.cuda.synchronize()
    assert torch.equal(v, v2) and torch.equal(i, i2) and i.dtype == torch.int64 and v.shape == (rows, k)
    assert j2.dtype == torch.int32 and torch.equal(j2.long(), i2)
    if n % 4 == 0:       # the same selection with the row formed inside the launch (eamd_weighted_topk_rows): bit-equal to sum, then top-k
        parts = [torch.randn(rows, n, generator=g).to(DEV) for _ in range(3)]
        wts = [0.7, 0.1, -0.35]
        pre_ref = ops.weighted_sum(parts, wts)
        vr, ir = ops.topk_rows(pre_ref, k)
        pre, i3, j3 = ops.weighted_topk_rows(parts, wts, k)
        torch.cuda.synchronize()
        assert torch.equal(pre, pre_ref) and torch.equal(i3, ir) and torch.equal(j3.long(), ir)
    xr = torch.where(torch.isnan(x), torch.full_like(x, -float("inf")), x)
    rv, _ = torch.topk(xr, k, dim=1)
    assert torch.equal(v.cpu(), rv)
    gathered = torch.gather(xr, 1, i.cpu())
    assert torch.equal(gathered, rv)                     # the indices address those values
    ic = i.cpu()
    for r in range(rows):                                # a permutation-free selection, ties by ascending index
        assert len(set(ic[r].tolist())) == k
        for a in range(k - 1):
            if rv[r, a] == rv[r, a + 1]:
                assert ic[r, a] < ic[r, a + 1]
    if rows > 2:
        assert ic[1].tolist() == list(range(k))


@pytest.mark.parametrize("M,N,K,a_act,act,res", [(10, 256, 256, 0, 0, True), (10, 2048, 256, 0, 1, False), (10, 256, 2048, 1, 0, True),
                                                 (16, 5000, 256, 0, 0, False), (1, 30, 64, 2, 2, True), (7, 257, 1028, 0, 0, True),
                                                 (320, 256, 256, 0, 0, True), (320, 256, 2048, 1, 0, True), (41, 2049, 256, 0, 1, False),
                                                 (17, 30, 64, 2, 2, True), (16, 255, 4096, 0, 2, True), (9, 64, 1024, 2, 0, False),
                                                 (12, 130, 3000, 0, 1, True), (33, 48, 1040, 0, 2, True), (1000, 256, 4096, 2, 0, False)])
def test_linear_rows_f32(ops, M, N, K, a_act, act, res):
    """eamd_linear_rows_f32 (nn.Linear on <= 16 rows, one wave per output column; taken by ops.linear_fwd without autograd) against
    float64: y = alpha * act(a_act(x) W^T + b) + R.  reference: the per-step products of decoder_layer.py:77-134."""
    import espnet_amd
    espnet_amd.set_precision("fp32")
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g) if res else None
    f = {0: lambda v: v, 1: lambda v: v.clamp_min(0), 2: lambda v: v * torch.sigmoid(v)}
    ref = 0.5 * f[act](f[a_act](x.double()) @ W.double().t() + b.double()) + (R.double() if res else 0.0)
    with torch.no_grad():
        keep_max, ops.LINEAR_ROWS_MAX = ops.LINEAR_ROWS_MAX, 1024          # also the row-block form for every shape (M > 16: by default only K <= 512, N <= 1024)
        try:
            y = ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), act=act, a_act=a_act, alpha=0.5, R=R.to(DEV) if res else None)
        finally:
            ops.LINEAR_ROWS_MAX = keep_max
        keep, ops.LINEAR_ROWS = ops.LINEAR_ROWS, False
        try:
            y_tiles = ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), act=act, a_act=a_act, alpha=0.5, R=R.to(DEV) if res else None)
        finally:
            ops.LINEAR_ROWS = keep
    report("linear_rows %dx%dx%d a_act=%d act=%d" % (M, N, K, a_act, act), y, ref, 2e-6)
    report("linear_rows vs the tile kernel", y, y_tiles, 2e-6)


@pytest.mark.parametrize("B,beam,V,nf,ctc,full_mode,pre", [(1, 10, 5000, 1, True, True, True), (3, 4, 30, 3, True, False, True),
                                                           (2, 5, 50, 2, True, False, False), (4, 3, 40, 1, False, False, False),
                                                           (2, 4, 30, 2, True, True, False)])
def test_beam_finish(ops, B, beam, V, nf, ctc, full_mode, pre):
    """eamd_beam_finish against the tensor expressions it replaced (beam_search.py:177-203 / batch_beam_search.py:249-284 on
    device tensors): hypothesis / token of each winner, carried scores, prefixes, running scores with ended / empty slots at
    -inf, position among the candidates, and the log row."""
    g = torch.Generator().manual_seed(B * 100 + V)
    n, W, L, step, eos = B * beam, 9, 3, 2, V - 1
    ncand = 7 if pre else V
    top_i = torch.stack([torch.randperm(beam * V, generator=g)[:beam] for _ in range(B)])           # distinct winners per utterance
    top_i[0, 0] = (top_i[0, 0] // V) * V + eos                                                      # one winner ends with <eos>
    top_s = torch.randn(B, beam, generator=g)
    top_s[-1, -1] = -float("inf")                                                                   # an empty slot
    maxlen = torch.tensor([step + 1] + [step + 5] * (B - 1))                                        # utterance 0 is at its length cap
    ns = nf + int(ctc)
    sc_in = torch.randn(ns, n, generator=g)
    logps = [torch.randn(n, V, generator=g) for _ in range(nf)]
    ids = torch.stack([torch.randperm(V, generator=g)[:ncand] for _ in range(n)]) if pre else None
    c_local = torch.randn(n, V if (full_mode or not pre) else ncand, generator=g) if ctc else None
    yseq_in = torch.randint(0, V, (n, W), generator=g)
    dv = lambda t: None if t is None else t.to(DEV)                                                 # noqa: E731
    out = ops.beam_finish(dv(top_s.reshape(-1)), dv(top_i.reshape(-1)), beam, V, L, step, eos, dv(maxlen), dv(sc_in),
                          [dv(lp) for lp in logps], dv(c_local), full_mode, dv(ids), dv(yseq_in))
    sc_out, yseq_out, hyp_out, hyp_i, tok_i, pos, rec = [t.cpu() for t in out]
    base = (torch.arange(B) * beam).view(B, 1)
    r_hyp = (top_i // V + base).view(-1)
    r_tok = (top_i % V).view(-1)
    assert torch.equal(hyp_i, r_hyp) and torch.equal(tok_i, r_tok)
    r_pos = (ids[r_hyp] == r_tok[:, None]).float().argmax(-1) if pre else r_tok
    assert torch.equal(pos, r_pos)
    for j in range(nf):
        assert torch.equal(sc_out[j], sc_in[j][r_hyp] + logps[j][r_hyp, r_tok])
    if ctc:
        col = r_tok if (full_mode or not pre) else r_pos
        assert torch.equal(sc_out[nf], sc_in[nf][r_hyp] + c_local[r_hyp, col])
    r_y = yseq_in.index_select(0, r_hyp).clone()
    r_y[:, L] = r_tok
    assert torch.equal(yseq_out, r_y)
    ts = top_s.reshape(-1)
    finite = torch.isfinite(ts)
    at_cap = (maxlen.view(B, 1) <= step + 1).expand(B, beam).reshape(-1)
    done = finite & ((r_tok == eos) | at_cap)
    r_hypo = torch.where(done | ~finite, torch.full_like(ts, -float("inf")), ts)
    assert torch.equal(hyp_out, r_hypo) and bool(done[0]) and float(hyp_out[-1]) == -float("inf")
    r_rec = torch.cat([torch.full((n, 1), float(step)), ts[:, None], r_tok[:, None].float()] + [sc_out[j][:, None] for j in range(ns)]
                      + [r_y.float()], dim=1)
    assert torch.equal(rec, r_rec)


@pytest.mark.parametrize("lens", [[249, 159, 74], [700, 513, 90], [1500, 1100, 1025]])
def test_ctc_prefix_score_vs_float64(ops, lens):
    """eamd_ctc_prefix_score and eamd_ctc_prefix_score_batch at the benchmarked size (T' = 249, |V| = 5000, 10 hypotheses x 15
    candidates, three utterances of 249 / 159 / 74 frames) against the float64 restatement of CTCPrefixScore.__call__
    (ctc_prefix_score.py:255-310; oracle.ctc_prefix_step) over five chained search steps - every step's input state is the
    float64 state of the chosen candidate, so rounding does not compound across steps and the bound is that of ONE 249-frame
    recursion: candidates include <eos>, blank and the prefix's last token (the r^b-only branch), frames whose posterior is
    -inf for a candidate AND for blank (np.logaddexp(-inf, -inf) = -inf), prefixes longer than a short utterance is not needed.
    Bound: |err| <= 1e-5 + 2e-6 |ref| (fp32 ulp at |r| ~ 300 is 3e-5; the float32 numpy arithmetic of the reference itself
    is measured beside it).  Longer utterances (700 and 1500 frames: 16 / 32 frames per lane in eamd_ctc_prefix_psi and in the
    scan of eamd_ctc_prefix_state) under the same bound."""
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asr_oracle as O
    V, per, P, blank, eos = (5000, 10, 15, 0, 4999) if max(lens) < 1000 else (600, 4, 9, 0, 599)      # (the long case: a smaller table)
    U, Tmax = len(lens), max(lens)
    g = torch.Generator().manual_seed(249)
    logits = 2.5 * torch.randn(U, Tmax, V, generator=g)
    logits[:, :, blank] += 9.0
    logp = torch.log_softmax(logits, -1)
    ninf_tok = 1234 if V > 1234 else 123
    logp[0, 40:43, ninf_tok] = -float("inf")
    logp[0, 41, blank] = -float("inf")                    # frame 41 of utterance 0: candidate 1234 and blank both impossible
    logp[1, 100, 77] = -float("inf")
    logp_d = logp.to(DEV).contiguous()
    lens_d = torch.tensor(lens, dtype=torch.int32, device=DEV)
    # float64 chain
    r_prev = np.stack([np.repeat(O.ctc_prefix_init(logp[u, :lens[u]].double().numpy(), blank, np.float64)[None], per, 0)
                       if lens[u] == Tmax else
                       np.concatenate([np.repeat(O.ctc_prefix_init(logp[u, :lens[u]].double().numpy(), blank, np.float64)[None], per, 0),
                                       np.zeros((per, Tmax - lens[u], 2))], 1) for u in range(U)])          # [U, per, Tmax, 2]
    last = np.full((U, per), eos, dtype=np.int64)
    worst = dict(psi=0.0, r=0.0, psi32=0.0, r32=0.0, rs=0.0)
    for step in range(5):
        cand = torch.randint(1, V - 1, (U, per, P), generator=g).numpy()
        cand[:, :, 0], cand[:, :, 1] = eos, blank
        if step > 0:
            cand[:, :, 2] = last                                                  # the repeated-label branch
        cand[0, :, 3] = ninf_tok
        cand[1, :, 3] = 77
        psi64, r64, psi32, r32 = [], [], [], []
        for u in range(U):
            T = lens[u]
            lp = logp[u, :T].double().numpy()
            p, r = O.ctc_prefix_step(lp, last[u], step, cand[u], r_prev[u][:, :T], blank, eos, np.float64)
            psi64.append(p); r64.append(r)
            with np.errstate(all="ignore"):
                p3, r3 = O.ctc_prefix_step(lp.astype(np.float32), last[u], step, cand[u], r_prev[u][:, :T].astype(np.float32), blank, eos, np.float32)
            psi32.append(p3); r32.append(r3)
        cand_d = torch.from_numpy(cand.reshape(U * per, P)).to(torch.int32).to(DEV)
        last_d = torch.from_numpy(last.reshape(-1)).to(torch.int32).to(DEV)
        olen_d = torch.full((U * per,), step, dtype=torch.int32, device=DEV)
        rp_d = torch.from_numpy(r_prev.reshape(U * per, Tmax, 2)).float().to(DEV)
        psi_b, r_b = ops.ctc_prefix_score_batch(logp_d, lens_d, per, rp_d, cand_d, last_d, olen_d, blank, eos)
        psi_1, r_1 = ops.ctc_prefix_score(logp_d[0], rp_d[:per].contiguous(), cand_d[:per].contiguous(), last_d[:per].contiguous(),
                                          olen_d[:per].contiguous(), blank, eos)
        torch.cuda.synchronize()
        assert torch.equal(psi_1, psi_b[:per]) and torch.equal(r_1, r_b[:per])    # one utterance == its rows of the batched launch

        def chk(name, got, ref, key):
            got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
            assert not np.isnan(got).any(), name
            inf = np.isinf(ref)
            assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), name + ": -inf pattern"
            e = np.abs(got[~inf] - ref[~inf]) / (1e-5 + 2e-6 * np.abs(ref[~inf]))
            worst[key] = max(worst[key], float(e.max()) if e.size else 0.0)
        for u in range(U):
            T = lens[u]
            sl = slice(u * per, (u + 1) * per)
            # rows the reference leaves uninitialised (before start - 1) are never read by a later step: excluded
            t0 = max(step, 1) - 1
            chk("psi u%d step%d" % (u, step), psi_b[sl].cpu().numpy(), psi64[u], "psi")
            chk("r u%d step%d" % (u, step), r_b[sl, :, t0:T].cpu().numpy(), r64[u][:, :, t0:], "r")
            chk("psi32", psi32[u], psi64[u], "psi32")
            chk("r32", r32[u][:, :, t0:], r64[u][:, :, t0:], "r32")
            assert float(r_b[sl, :, T:].abs().max()) == 0.0 if T < Tmax else True   # rows beyond the utterance stay zero
        # next step: each hypothesis takes one of its candidates (not blank / <eos>), state = the float64 one
        pick = torch.randint(2, P, (U, per), generator=g).numpy()
        # the split form of a beam step: candidates scored by the parallel reduction (eamd_ctc_prefix_psi: same bound against
        # float64), the survivors' forward variables by eamd_ctc_prefix_state: bit-equal to the full recursion's rows
        psi_p = ops.ctc_prefix_psi(logp_d, lens_d, per, rp_d, cand_d, last_d, step, blank, eos)
        for u in range(U):
            chk("psi (parallel) u%d step%d" % (u, step), psi_p[u * per:(u + 1) * per].cpu().numpy(), psi64[u], "psi")
        tok_d = torch.from_numpy(np.take_along_axis(cand.reshape(U * per, P), pick.reshape(U * per, 1), 1)[:, 0].copy()).to(DEV)
        alive = torch.zeros(U * per, device=DEV)
        alive[3] = -float("inf")
        r_s = ops.ctc_prefix_state(logp_d, lens_d, per, rp_d, torch.arange(U * per, device=DEV), tok_d, last_d, step, alive, blank)
        torch.cuda.synchronize()
        t0 = max(step, 1) - 1
        for u in range(U):
            for hh in range(per):
                s_ = u * per + hh
                if s_ == 3:
                    assert float(r_s[s_, :lens[u]].max()) == -10000000000.0
                else:      # (a parallel scan over the frames: same bound against float64 as the frame-by-frame recursion)
                    chk("r (state) u%d step%d" % (u, step), r_s[s_, t0:lens[u]].cpu().numpy(), r64[u][hh, int(pick[u, hh]), t0:], "rs")
                    assert float(r_s[s_, :t0].max()) <= -9.9e9 if t0 > 0 else True
        for u in range(U):
            T = lens[u]
            for h in range(per):
                r_prev[u, h, :T] = r64[u][h, pick[u, h]]
                last[u, h] = cand[u, h, pick[u, h]]
    print("[parity] ctc_prefix_score vs float64, worst err / (1e-5 + 2e-6 |ref|): HIP psi %.3f r %.3f | reference float32 "
          "arithmetic (numpy) psi %.3f r %.3f | survivors' states by the scan %.3f" % (worst["psi"], worst["r"], worst["psi32"], worst["r32"], worst["rs"]))
    # (the frame-by-frame recursion compounds its rounding over the frames: at 1500 frames the reference's own float32 arithmetic
    # is at 1.6 of the bound set for 249 - the recursion kernel may be where that arithmetic is, not beyond; the scan stays inside)
    assert worst["psi"] <= 1.0 and worst["r"] <= max(1.0, 1.05 * worst["r32"]) and worst["rs"] <= 1.0


@pytest.mark.parametrize("M", [7968, 100, 32])
def test_rowproj_vs_float64(ops, M):
    """eamd_rowproj (32 rows per workgroup through the whole product, packed weight images) against float64:
    y = x W^T + b for (K, N) = (256, 768 / 512 / 256) and dx = dy W for K = 768 / 512 / 256 -> 256, row-strided operands, bias,
    alpha, residual; output dropout = eamd_dropout's mask on the plain product; LayerNorm in front (rows, mean, rstd as
    eamd_layernorm_fwd); BatchNorm-style affine + Swish on the staged rows; the LayerNorm BACKWARD behind an input gradient
    (dx, residual gradient, dropped copy, d gamma / d beta partials) against eamd_layernorm_bwd on the GEMM's result."""
    g = torch.Generator().manual_seed(M)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    for K, N in ((256, 768), (256, 512), (256, 256), (768, 256), (512, 256)):
        for trans in (False, True):
            W = rnd(K, N) / K ** 0.5 if trans else rnd(N, K) / K ** 0.5
            x, b, R = rnd(M, K), rnd(N), rnd(M, N)
            Wd, xd, bd, Rd = W.to(DEV), x.to(DEV), b.to(DEV), R.to(DEV)
            img, = ops.rowproj_pack([(Wd, trans)])
            Bm = W.double() if trans else W.double().t()
            y = ops.rowproj(xd, img, N, bias=bd, R=Rd, alpha=0.5)
            report(f"rowproj M={M} K={K} N={N} trans={int(trans)}", y, R.double() + 0.5 * (x.double() @ Bm + b.double()), 2e-6)
            # row-strided input / output (column blocks of wider buffers, as the q / k / v blocks of a fused projection are)
            wide_in = torch.zeros(M, K + 64, device=DEV)
            wide_in[:, 32:32 + K] = xd
            wide_out = torch.full((M, N + 128), 7.0, device=DEV)
            ops.rowproj(wide_in[:, 32:32 + K], img, N, out=wide_out[:, 64:64 + N])
            report("rowproj strided", wide_out[:, 64:64 + N], x.double() @ Bm, 2e-6)
            assert float(wide_out[:, :64].min()) == 7.0 and float(wide_out[:, 64 + N:].max()) == 7.0
    # dropout + residual epilogue: the mask eamd_dropout draws for the contiguous [M, N] product
    K, N = 256, 256
    W, x, b, R = rnd(N, K) / 16, rnd(M, K), rnd(N), rnd(M, N)
    Wd, xd, bd, Rd = W.to(DEV), x.to(DEV), b.to(DEV), R.to(DEV)
    img, = ops.rowproj_pack([(Wd, False)])
    ops.manual_seed(77)
    plain = ops.rowproj(xd, img, N, bias=bd)
    want = Rd + ops.dropout(plain, 0.1, 4242)
    got = ops.rowproj(xd, img, N, bias=bd, R=Rd, drop=(0.1, 4242))
    torch.cuda.synchronize()
    assert torch.equal((got - Rd == 0), (want - Rd == 0)) or float(((got - Rd == 0) != (want - Rd == 0)).float().mean()) < 1e-6
    report("rowproj dropout + residual", got, want, 1e-6)
    # LayerNorm in front
    gam, bet = (1.0 + 0.1 * rnd(K)).to(DEV), (0.1 * rnd(K)).to(DEV)
    xn_ref, mean_ref, rstd_ref = ops.layernorm_fwd(xd, gam, bet, 1e-12, torch.float32)
    xn = torch.empty(M, K, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    y = ops.rowproj(xn, img, N, bias=bd, ln=(xd, gam, bet, 1e-12, mean, rstd))
    report("rowproj LN rows", xn, xn_ref, 1e-6)
    report("rowproj LN mean", mean, mean_ref[:M], 1e-6)
    report("rowproj LN rstd", rstd, rstd_ref[:M], 1e-6)
    ln64 = torch.nn.functional.layer_norm(x.double(), (K,), gam.cpu().double(), bet.cpu().double(), 1e-12)
    report("rowproj LN + product", y, ln64 @ W.double().t() + b.double(), 2e-6)
    # affine + Swish on the staged rows (BatchNorm apply in front of pointwise_conv2)
    sc, sh = (1.0 + 0.2 * rnd(K)).to(DEV), (0.3 * rnd(K)).to(DEV)
    e_out = torch.empty(M, K, device=DEV)
    y = ops.rowproj(xd, img, N, bias=bd, affine=(sc, sh, ops.ACT_SWISH, e_out))
    z = x.double() * sc.cpu().double() + sh.cpu().double()
    e64 = z * torch.sigmoid(z)
    report("rowproj affine rows", e_out, e64, 2e-6)
    report("rowproj affine + product", y, e64 @ W.double().t() + b.double(), 3e-6)
    # LayerNorm backward behind the input gradient dxn = dy W (K = 768 -> 256)
    Kb = 768
    W3, dy = rnd(Kb, 256) / Kb ** 0.5, rnd(M, Kb)
    x_in, dres = rnd(M, 256), rnd(M, 256)
    W3d, dyd, xind, dresd = W3.to(DEV), dy.to(DEV), x_in.to(DEV), dres.to(DEV)
    imgb, = ops.rowproj_pack([(W3d, True)])
    _, mean_i, rstd_i = ops.layernorm_fwd(xind, gam, bet, 1e-12, torch.float32)
    dxn = ops.rowproj(dyd, imgb, 256)
    dg_ref, db_ref = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    ops.manual_seed(78)
    dx_ref, dxd_ref = ops.layernorm_bwd(dxn, xind, gam, mean_i, rstd_i, dresd, dg_ref, db_ref, drop=(0.1, 999))
    ws = ops.rowproj_lnb_ws(M, DEV)
    ddrop = torch.empty(M, 256, device=DEV)
    dx = ops.rowproj(dyd, imgb, 256, lnb=(xind, gam, mean_i, rstd_i, dresd, ws, ddrop, (0.1, 999)))
    torch.cuda.synchronize()
    report("rowproj LN-bwd dx", dx, dx_ref, 2e-6)
    report("rowproj LN-bwd dropped copy", ddrop, dxd_ref.float(), 2e-6)
    part = ws.view(-1, 2, 256).sum(0)
    report("rowproj LN-bwd d gamma", part[0], dg_ref, 2e-5)
    report("rowproj LN-bwd d beta", part[1], db_ref, 2e-5)
    # float64 check of the whole chain on a few rows
    xi = x_in[:8].double().requires_grad_(True)
    gam64 = gam.cpu().double()
    yl = torch.nn.functional.layer_norm(xi, (256,), gam64, bet.cpu().double(), 1e-12)
    (yl * (dy[:8].double() @ W3.double())).sum().backward()
    report("rowproj LN-bwd vs float64", dx[:8], xi.grad + dres[:8].double(), 5e-6)


def test_ffn_bwd_layernorm_backward_epilogue(ops):
    """eamd_ffn_bwd with the block's LayerNorm backward as its epilogue (eamd_ffn_t.lnb_*): dz unchanged, dx / dropped copy /
    d gamma / d beta as eamd_layernorm_bwd gives them on the plain launch's dx; ragged last row block."""
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    F = 512
    for M in (4128, 100):
        w1, w2 = rnd(F, 256) / 16, rnd(256, F) / F ** 0.5
        dy, f, x_in, dres = rnd(M, 256), rnd(M, F), rnd(M, 256), rnd(M, 256)
        gam, bet = 1.0 + 0.1 * rnd(256), 0.1 * rnd(256)
        _, mean, rstd = ops.layernorm_fwd(x_in, gam, bet, 1e-12, torch.float32)
        old = ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_HSPLIT_MIN_ROWS
        ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_HSPLIT_MIN_ROWS = 1, 1 << 30
        try:
            dz0, dxn = ops.ffn_bwd(dy, w1, w2, f, alpha=0.5)
            dg_ref, db_ref = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
            ops.manual_seed(9)
            dx_ref, dxd_ref = ops.layernorm_bwd(dxn, x_in, gam, mean, rstd, dres, dg_ref, db_ref, drop=(0.1, 31))
            ws = ops.rowproj_lnb_ws(M, DEV)
            dcopy = torch.empty(M, 256, device=DEV)
            dz1, dx = ops.ffn_bwd(dy, w1, w2, f, alpha=0.5, lnb=(x_in, gam, mean, rstd, dres, ws, dcopy, (0.1, 31)))
        finally:
            ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_HSPLIT_MIN_ROWS = old
        torch.cuda.synchronize()
        assert torch.equal(dz0, dz1)
        report("ffn_bwd + LN-bwd dx (M=%d)" % M, dx, dx_ref, 2e-6)
        report("ffn_bwd + LN-bwd dropped copy", dcopy, dxd_ref.float(), 2e-6)
        part = ws.view(-1, 2, 256).sum(0)
        report("ffn_bwd + LN-bwd d gamma", part[0], dg_ref, 2e-5)
        report("ffn_bwd + LN-bwd d beta", part[1], db_ref, 2e-5)


def test_decode_kernels_vs_float64(ops):
    """csrc/decode.hip: eamd_linear_rows_ln_f32 against float64 LayerNorm + Linear (+ ReLU, residual, strided rows); the cached
    self-attention (eamd_decode_self_attn + eamd_beam_slots) over six beam steps with random re-ordering of the hypotheses
    against softmax attention in float64 over each hypothesis's TRUE history (keys / values gathered along its ancestry)."""
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    for M, K, N in ((10, 256, 768), (1, 256, 256), (16, 256, 5000), (7, 512, 130)):
        x, W, b, R = rnd(M, K) * 2 + 0.5, rnd(N, K) / K ** 0.5, rnd(N), rnd(M, N)
        gam, bet = 1 + 0.1 * rnd(K), 0.1 * rnd(K)
        ln = torch.nn.functional.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-12)
        y = ops.linear_rows_ln(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-12, W.to(DEV), b.to(DEV), act=ops.EPI_RELU, R=R.to(DEV), alpha=0.5)
        assert y is not None
        report(f"linear_rows_ln {M}x{K}x{N}", y, 0.5 * torch.relu(ln @ W.double().t() + b.double()) + R.double(), 2e-6)
        wide = torch.zeros(M, 3, K)
        wide[:, 2] = x
        y2 = ops.linear_rows_ln(wide.to(DEV)[:, 2], gam.to(DEV), bet.to(DEV), 1e-12, W.to(DEV), None)
        report("linear_rows_ln strided rows", y2, ln @ W.double().t(), 2e-6)
    assert ops.linear_rows_ln(rnd(17, 512).to(DEV), rnd(512).to(DEV), rnd(512).to(DEV), 1e-12, rnd(8, 512).to(DEV), None) is None
    # 16-row blocks on the matrix cores (a batched search's utterances x beam rows), K <= 256
    for (M, K, N) in ((320, 256, 768), (33, 256, 1024), (17, 64, 30), (1000, 256, 256)):
        x, gam, bet, W, b, R = rnd(M, K), 1.0 + 0.1 * rnd(K), 0.1 * rnd(K), rnd(N, K) / K ** 0.5, 0.1 * rnd(N), rnd(M, N)
        y = ops.linear_rows_ln(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-12, W.to(DEV), b.to(DEV), act=ops.EPI_RELU, R=R.to(DEV), alpha=0.5)
        ln = torch.nn.functional.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-12)
        report(f"linear_rows_ln (16-row blocks) {M}x{K}x{N}", y, 0.5 * torch.relu(ln @ W.double().t() + b.double()) + R.double(), 2e-6)
    n, H, D, Lcap = 10, 4, 256, 24
    Kc, Vc = torch.full((Lcap, n, D), float("nan"), device=DEV), torch.full((Lcap, n, D), float("nan"), device=DEV)
    slot = torch.zeros(n, Lcap, dtype=torch.int32, device=DEV)
    hist_k = [[] for _ in range(n)]          # per current slot: list of (k, v) rows of its true history
    for pos in range(6):
        qkv = rnd(n, 3 * D)
        ctx = ops.decode_self_attn(qkv.to(DEV), Kc, Vc, slot, pos, H)
        for i in range(n):
            hist_k[i] = hist_k[i] + [(qkv[i, D:2 * D].double(), qkv[i, 2 * D:].double())]
        want = torch.zeros(n, D, dtype=torch.float64)
        for i in range(n):
            Kh = torch.stack([kv[0] for kv in hist_k[i]]).view(-1, H, 64)
            Vh = torch.stack([kv[1] for kv in hist_k[i]]).view(-1, H, 64)
            q = qkv[i, :D].double().view(H, 64)
            p = torch.softmax(torch.einsum("hd,thd->ht", q, Kh) / 8.0, -1)
            want[i] = torch.einsum("ht,thd->hd", p, Vh).reshape(D)
        report(f"decode_self_attn step {pos}", ctx, want, 2e-6)
        hyp = torch.randint(0, n, (n,), generator=g)            # the selection: slot i continues the hypothesis of slot hyp[i]
        slot = ops.beam_slots(slot, hyp.to(DEV), pos)
        hist_k = [list(hist_k[int(h)]) for h in hyp]
        assert slot[:, pos].cpu().tolist() == hyp.tolist()


def test_decode_src_attn_vs_float64(ops):
    """eamd_decode_src_attn: one query position per hypothesis over the memory of its utterance (keys / values as column blocks of
    a wider projection buffer, padded frames masked, one fully masked utterance) against float64 softmax attention"""
    g = torch.Generator().manual_seed(21)
    G, gb, T, H, D, L = 3, 5, 77, 4, 256, 2
    kv = torch.randn(G * T, 2 * D * L, generator=g)             # [rows, layers x (k | v)]
    q = torch.randn(G * gb, D, generator=g)
    lens = [77, 40, 0]
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).to(torch.uint8).unsqueeze(1).contiguous()
    for layer in range(L):
        k_off, v_off = layer * 2 * D, layer * 2 * D + D
        ctx = ops.decode_src_attn(q.to(DEV), kv.to(DEV), k_off, v_off, 2 * D * L, mask.to(DEV), G, gb, T, H)
        want = torch.zeros(G * gb, D, dtype=torch.float64)
        for r in range(G * gb):
            u = r // gb
            if lens[u] == 0:
                continue
            K = kv[u * T: u * T + lens[u], k_off:k_off + D].double().view(-1, H, 64)
            Vv = kv[u * T: u * T + lens[u], v_off:v_off + D].double().view(-1, H, 64)
            p = torch.softmax(torch.einsum("hd,thd->ht", q[r].double().view(H, 64), K) / 8.0, -1)
            want[r] = torch.einsum("ht,thd->hd", p, Vv).reshape(D)
        report("decode_src_attn layer %d" % layer, ctx, want, 2e-6)
        assert float(ctx[2 * gb:].abs().max()) == 0.0
        # ... one workgroup per (utterance, head) for all hypotheses of the utterance
        ctx_g = ops.decode_src_attn(q.to(DEV), kv.to(DEV), k_off, v_off, 2 * D * L, mask.to(DEV), G, gb, T, H, group=True)
        report("decode_src_attn (grouped) layer %d" % layer, ctx_g, want, 2e-6)
        assert float(ctx_g[2 * gb:].abs().max()) == 0.0
    ctx = ops.decode_src_attn(q.to(DEV), kv.to(DEV), 0, D, 2 * D * L, None, G, gb, T, H)
    K = kv[:T, :D].double().view(-1, H, 64)
    p = torch.softmax(torch.einsum("hd,thd->ht", q[0].double().view(H, 64), K) / 8.0, -1)
    report("decode_src_attn no mask", ctx[0], torch.einsum("ht,thd->hd", p, kv[:T, D:2 * D].double().view(-1, H, 64)).reshape(D), 2e-6)
    # the benchmarked shape of a batched search: 32 utterances x 10 hypotheses, 249 frames
    G, gb, T = 32, 10, 249
    kv = torch.randn(G * T, 2 * D, generator=g)
    q = torch.randn(G * gb, D, generator=g)
    ctx_g = ops.decode_src_attn(q.to(DEV), kv.to(DEV), 0, D, 2 * D, None, G, gb, T, H, group=True)
    K = kv[:, :D].double().view(G, T, H, 64)
    Vv = kv[:, D:].double().view(G, T, H, 64)
    p = torch.softmax(torch.einsum("gbhd,gthd->gbht", q.double().view(G, gb, H, 64), K) / 8.0, -1)
    report("decode_src_attn (grouped) 32 x 10 x 249", ctx_g, torch.einsum("gbht,gthd->gbhd", p, Vv).reshape(G * gb, D), 2e-6)
    assert ops.decode_src_attn(q.to(DEV), kv.to(DEV), 0, D, 2 * D, None, 16, 20, T, H, group=True) is None       # more than 16 per utterance


def test_graph_audit_rejects_memset_nodes():
    """espnet_amd.graphs.audit: a captured graph that holds a MEMSET node is refused (on this ROCm such a node replays wrongly from
    the second launch on: profiles/r04_graph_census.txt, DESIGN.md section 2 "Round 4").  torch.topk's multi-block path ([320, 5000])
    zeroes its counters with memsets - the op whose step graphs faulted in round 3; its single-block path and our own selection
    kernel capture kernels only.  Nothing captured here is ever replayed."""
    from espnet_amd import _lib, graphs, ops as o
    x320 = torch.randn(320, 5000, device=DEV)
    x10 = torch.randn(10, 5000, device=DEV)

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = graphs.new_graph()
        with torch.cuda.graph(g):
            fn()
        return g

    kinds, memsets = graphs.node_census(capture(lambda: torch.topk(x320, 10, dim=-1)))
    print("[parity] torch.topk [320, 5000] captured as", kinds, memsets)
    assert kinds.get("memset", 0) >= 1 and len(memsets) == kinds["memset"]
    with pytest.raises(_lib.EamdError):
        graphs.audit(capture(lambda: torch.topk(x320, 10, dim=-1)), "a graph with torch.topk's multi-block path")
    assert "memset" not in graphs.audit(capture(lambda: torch.topk(x10, 15, dim=-1)))
    assert graphs.audit(capture(lambda: o.topk_rows(x320, 10))) == {"kernel": 1}
    assert graphs.capture_id() == 0
