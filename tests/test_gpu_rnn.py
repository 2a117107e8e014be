"""GPU parity tests of the RNN paths: HIP kernels behind the C ABI
against the CPU oracle on seeded inputs and against vectors recorded from the reference (tests/golden).
fp32 tolerances are written at each check; integer work (lengths, pooling indices) is exact."""
import argparse

import numpy as np
import pytest
import torch

from conftest import load_golden, split_golden
from test_gpu_model import check_grads, load_sd
from test_gpu_ops import rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _fp32():
    import espnet_amd
    espnet_amd.set_precision("fp32")
    yield
    espnet_amd.set_precision("fp32")


# ---- kernels ---------------------------------------------------------------------------------------
def test_maxpool_ceil_mode():
    from espnet_amd import ops
    g = torch.Generator().manual_seed(0)
    for (B, H, W, C) in ((2, 7, 5, 3), (1, 8, 6, 64), (3, 1, 9, 5)):
        x = torch.randn(B, H, W, C, generator=g)
        want = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 2, stride=2, ceil_mode=True).permute(0, 2, 3, 1)
        y, idx = ops.maxpool2x2_fwd(x.to(DEV))
        assert torch.equal(y.cpu(), want)                              # selection: bit-exact
        gy = torch.randn(want.shape, generator=g)
        xr = x.clone().requires_grad_(True)
        torch.nn.functional.max_pool2d(xr.permute(0, 3, 1, 2), 2, stride=2, ceil_mode=True).permute(0, 2, 3, 1) \
            .backward(gy)
        dx = ops.maxpool2x2_bwd(gy.to(DEV), idx, (B, H, W, C))
        assert torch.equal(dx.cpu(), xr.grad)


@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_sequence_vs_oracle(oracle, reverse):
    from espnet_amd import rnn_functional as R
    g = torch.Generator().manual_seed(3)
    T, B, I, H = 9, 4, 6, 5
    lens = [9, 7, 4, 1]
    sd = {"weight_ih_l0": torch.randn(4 * H, I, generator=g) * 0.4, "weight_hh_l0": torch.randn(4 * H, H, generator=g) * 0.4,
          "bias_ih_l0": torch.randn(4 * H, generator=g) * 0.2, "bias_hh_l0": torch.randn(4 * H, generator=g) * 0.2}
    x = torch.randn(B, T, I, generator=g)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    want = oracle.lstm_layer(sdr, "", xr, lens, reverse=reverse)
    gy = torch.randn(want.shape, generator=g)
    want.backward(gy)

    w_ih, w_hh, b_ih, b_hh = (sd[k].to(DEV).requires_grad_(True) for k in
                              ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"))
    xd = x.to(DEV).requires_grad_(True)
    from espnet_amd import functional as F_
    gx = F_.LinearFn.apply(xd.transpose(0, 1).contiguous(), w_ih, b_ih)
    live = torch.from_numpy((np.arange(T)[:, None] < np.asarray(lens)[None, :]).astype(np.uint8)).to(DEV)
    y = R.LSTMSeqFn.apply(gx, w_hh, b_hh, live, reverse).transpose(0, 1)
    report("lstm seq fwd rev=%d" % reverse, y, want.detach(), 1e-5)
    y.backward(gy.to(DEV))
    report("lstm seq dx", xd.grad, xr.grad, 1e-4)
    report("lstm seq dW_hh", w_hh.grad, sdr["weight_hh_l0"].grad, 1e-4)
    report("lstm seq db_hh", b_hh.grad, sdr["bias_hh_l0"].grad, 1e-4)
    report("lstm seq dW_ih", w_ih.grad, sdr["weight_ih_l0"].grad, 1e-4)


def test_lstm_cell_vs_oracle(oracle):
    from espnet_amd.nets.rnn.decoders import LSTMCell
    g = torch.Generator().manual_seed(4)
    B, I, H = 5, 7, 6
    cell = LSTMCell(I, H)
    sd = {k: v.detach().clone() for k, v in cell.state_dict().items()}
    cell = cell.to(DEV)
    x, h, c = torch.randn(B, I, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, hr, cr = (t.clone().requires_grad_(True) for t in (x, h, c))
    h2w, c2w = oracle.lstm_cell(sdr, "", xr, hr, cr)
    gh, gc = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    (h2w * gh).sum().add((c2w * gc).sum()).backward()
    xd, hd, cd = (t.to(DEV).requires_grad_(True) for t in (x, h, c))
    h2, c2 = cell(xd, (hd, cd))
    report("lstm cell h", h2, h2w.detach(), 1e-5)
    report("lstm cell c", c2, c2w.detach(), 1e-5)
    torch.autograd.backward([h2, c2], [gh.to(DEV), gc.to(DEV)])
    report("lstm cell dx", xd.grad, xr.grad, 1e-4)
    report("lstm cell dh", hd.grad, hr.grad, 1e-4)
    report("lstm cell dc", cd.grad, cr.grad, 1e-4)
    for k, p in cell.named_parameters():
        report("lstm cell d" + k, p.grad, sdr[k].grad, 1e-4)


@pytest.mark.parametrize("dims", [(3, 11, 6, 5, 7, 4, 2, [11, 8, 5]), (3, 45, 6, 5, 40, 10, 3, [45, 8, 5]),
                                  (2, 70, 12, 9, 1024, 10, 20, [70, 33]), (3, 45, 6, 5, 40, 4, 3, [45, 30, 5]),
                                  (2, 20, 5, 4, 16, 6, 2, [20, 7])])
def test_attloc_step_vs_oracle(oracle, dims):
    """two chained AttLoc steps (the second consumes the first's weights) against rnn/attentions.py:300-380.
    A = 7: the GEMM form of the backward products over mlp_att's weight; A = 40 / 1024: eamd_attloc_bwd_energy_conv"""
    from espnet_amd.nets.rnn.attentions import AttLoc
    g = torch.Generator().manual_seed(5)
    B, T, E, D, A, C, F, lens = dims
    att = AttLoc(E, D, A, C, F)
    sd = {k: v.detach().clone() for k, v in att.state_dict().items()}
    att = att.to(DEV)
    enc = torch.randn(B, T, E, generator=g)
    z1, z2 = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    encr, z1r, z2r = (t.clone().requires_grad_(True) for t in (enc, z1, z2))
    pre = oracle.linear(sdr, "mlp_enc.", encr)
    c1w, w1w = oracle.att_loc(sdr, "", encr, pre, lens, z1r, None)
    c2w, w2w = oracle.att_loc(sdr, "", encr, pre, lens, z2r, w1w)
    g1, g2, gw = torch.randn(B, E, generator=g), torch.randn(B, E, generator=g), torch.randn(B, T, generator=g)
    ((c1w * g1).sum() + (c2w * g2).sum() + (w2w * gw).sum()).backward()

    encd, z1d, z2d = (t.to(DEV).requires_grad_(True) for t in (enc, z1, z2))
    att.reset()
    c1, w1 = att(encd, lens, z1d, None)
    c2, w2 = att(encd, lens, z2d, w1)
    report("attloc ctx1", c1, c1w.detach(), 1e-5)
    report("attloc w1", w1, w1w.detach(), 1e-5)
    report("attloc ctx2", c2, c2w.detach(), 1e-5)
    assert float(w2[1, lens[1]:].abs().max()) == 0.0 and float(w2[B - 1, lens[-1]:].abs().max()) == 0.0   # masked frames: exactly 0
    torch.autograd.backward([c1, c2, w2], [g1.to(DEV), g2.to(DEV), gw.to(DEV)])
    report("attloc d enc_h", encd.grad, encr.grad, 1e-4)
    report("attloc d z1", z1d.grad, z1r.grad, 1e-4)
    report("attloc d z2", z2d.grad, z2r.grad, 1e-4)
    for k, p in att.named_parameters():
        if k == "gvec.bias":     # softmax is shift invariant: the exact gradient is 0, both sides hold rounding noise
            assert float(p.grad.abs().max()) < 1e-5
            continue
        report("attloc d" + k, p.grad, sdr[k].grad, 2e-4)


def test_attloc_fused_backward_matches_gemm_form():
    """eamd_attloc_bwd_energy_conv (df, dconv = df @ W_att, dW_att += df^T conv, dgvec, d dec from one pass over th, partial
    sums reduced without atomics) against the energy-backward kernel + two GEMMs, at config 4's attention size"""
    from espnet_amd import ops
    from espnet_amd.nets.rnn.attentions import AttLoc
    g = torch.Generator().manual_seed(11)
    B, T, E, D, A, C, F = 4, 250, 64, 48, 1024, 10, 100
    att = AttLoc(E, D, A, C, F).to(DEV)
    lens = [250, 201, 133, 64]
    enc = torch.randn(B, T, E, generator=g).to(DEV)
    z1, z2 = torch.randn(B, D, generator=g).to(DEV), torch.randn(B, D, generator=g).to(DEV)
    g1, g2, gw = (torch.randn(B, E, generator=g).to(DEV), torch.randn(B, E, generator=g).to(DEV),
                  torch.randn(B, T, generator=g).to(DEV))
    res = {}
    try:
        for fused in (True, False):
            ops.ATTLOC_FUSED_BWD = fused
            encd, z1d, z2d = (t.clone().requires_grad_(True) for t in (enc, z1, z2))
            att.zero_grad()
            att.reset()
            c1, w1 = att(encd, lens, z1d, None)
            c2, w2 = att(encd, lens, z2d, w1)
            torch.autograd.backward([c1, c2, w2], [g1, g2, gw])
            res[fused] = dict(enc=encd.grad.clone(), z1=z1d.grad.clone(), z2=z2d.grad.clone(),
                              **{k: p.grad.clone() for k, p in att.named_parameters()})
    finally:
        ops.ATTLOC_FUSED_BWD = True
    for k in res[True]:
        if k == "gvec.bias":
            continue
        report("attloc fused bwd " + k, res[True][k], res[False][k], 2e-4)


@pytest.mark.parametrize("shape", [(2, 6, 4, 7), (3, 17, 9, 33), (1, 1, 1, 5), (2, 5, 1, 4), (2, 1, 4, 6)])
def test_rnnt_loss_vs_oracle(oracle, shape):
    from espnet_amd import rnn_functional as R
    B, T, U, V = shape
    g = torch.Generator().manual_seed(B * 100 + T)
    z = torch.randn(B, T, U, V, generator=g) * 2.0
    y = torch.randint(1, V, (B, max(U - 1, 1)), generator=g)[:, : U - 1].int()
    tl = torch.tensor([T] + [max(1, T - 1 - i) for i in range(B - 1)], dtype=torch.int32)
    ul = torch.tensor([U - 1] + [max(0, U - 2 - i) for i in range(B - 1)], dtype=torch.int32)
    zr = z.clone().requires_grad_(True)
    want = oracle.rnnt_loss(zr, y, tl.tolist(), ul.tolist())
    want.backward()
    zd = z.to(DEV).requires_grad_(True)
    got = R.RNNTLossFn.apply(zd, y.to(DEV).contiguous(), tl.to(DEV), ul.to(DEV), 0)
    rel = abs(float(got) - float(want)) / abs(float(want))
    print("[parity] rnnt loss %s: hip %.6f oracle %.6f rel %.2e" % (shape, float(got), float(want), rel))
    assert rel < 1e-5
    got.backward()
    report("rnnt dlogits %s" % (shape,), zd.grad, zr.grad.float(), 2e-4)
    # rows outside each utterance's lattice receive exactly zero gradient
    for b in range(B):
        assert float(zd.grad[b, int(tl[b]):].abs().max() if int(tl[b]) < T else 0.0) == 0.0
        assert float(zd.grad[b, :, int(ul[b]) + 1:].abs().max() if int(ul[b]) + 1 < U else 0.0) == 0.0


def test_joint_network_vs_oracle(oracle):
    from espnet_amd.nets.transducer.joint_network import JointNetwork
    g = torch.Generator().manual_seed(8)
    for act in ("tanh", "relu", "swish"):
        jn = JointNetwork(9, 6, 5, 7, act)
        sd = {"joint_network." + k: v.detach().clone().requires_grad_(True) for k, v in jn.state_dict().items()}
        jn = jn.to(DEV)
        he, hd = torch.randn(2, 5, 6, generator=g), torch.randn(2, 4, 5, generator=g)
        her, hdr = he.clone().requires_grad_(True), hd.clone().requires_grad_(True)
        j = "joint_network."
        pre = oracle.linear(sd, j + "lin_enc.", her).unsqueeze(2) + \
            torch.nn.functional.linear(hdr, sd[j + "lin_dec.weight"]).unsqueeze(1)
        want = oracle.linear(sd, j + "lin_out.", oracle.activation(act)(pre))
        gz = torch.randn(want.shape, generator=g)
        want.backward(gz)
        hed, hdd = he.to(DEV).requires_grad_(True), hd.to(DEV).requires_grad_(True)
        z = jn(hed.unsqueeze(2), hdd.unsqueeze(1))
        report("joint %s fwd" % act, z, want.detach(), 1e-5)
        z.backward(gz.to(DEV))
        report("joint %s d h_enc" % act, hed.grad, her.grad, 1e-4)
        report("joint %s d h_dec" % act, hdd.grad, hdr.grad, 1e-4)
        for k, p in jn.named_parameters():
            report("joint %s d%s" % (act, k), p.grad, sd[j + k].grad, 1e-4)


# ---- models against the reference fixtures --------------------------------------------------------------
def _rnn_args(**kw):
    d = dict(elayers=2, subsample="1_2_1", etype="vggblstmp", eunits=12, eprojs=10, dtype="lstm", dlayers=2, dunits=14,
             atype="location", aheads=1, awin=3, aconv_chans=3, aconv_filts=2, mtlalpha=0.5, lsm_type="", lsm_weight=0.0,
             sampling_probability=0.0, adim=9, dropout_rate=0.0, dropout_rate_decoder=0.0, verbose=0,
             char_list=["<blank>", "a", "b", "c", "d", "e", "<eos>"], outdir=None, ctc_type="builtin",
             sym_space="<space>", sym_blank="<blank>", context_residual=False, use_frontend=False, replace_sos=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_e2e_rnn_golden():
    """VGG-BLSTMP + AttLoc LSTM decoder + CTC == reference E2E (e2e_asr.py:205-338) on its own weights"""
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn.npz"))
    m = load_sd(E2E(12, 7, _rnn_args()), sd)
    m.train()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    assert m.hlens == p["hlens"].tolist()
    report("e2e_rnn hs_pad", m.hs_pad, p["hs_pad"], 1e-4)
    for name, got, want in (("loss", loss, p["loss"]), ("loss_att", m.loss_att, p["loss_att"]),
                            ("loss_ctc", m.loss_ctc, p["loss_ctc"])):
        rel = abs(float(got) - float(want)) / abs(float(want))
        print("[parity] e2e_rnn %s hip %.6f ref %.6f rel %.2e" % (name, float(got), float(want), rel))
        assert rel < 1e-5
    assert abs(float(m.acc) - float(p["acc"])) < 1e-6
    loss.backward()
    check_grads(m, grads, tol=5e-4)


def test_e2e_rnn_scheduled_sampling_golden():
    """scheduled sampling (rnn/decoders.py:249-254, sampling_probability 0.5): with the Python `random` stream seeded as
    the fixture's run the decoder embeds its own argmax token at the same steps - loss, accuracy and every gradient equal
    the reference's"""
    import random
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn_ss.npz"))
    m = load_sd(E2E(12, 7, _rnn_args(sampling_probability=0.5)), sd)
    m.train()
    random.seed(7)
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    for name, got, want in (("loss", loss, p["loss"]), ("loss_att", m.loss_att, p["loss_att"]),
                            ("loss_ctc", m.loss_ctc, p["loss_ctc"])):
        rel = abs(float(got) - float(want)) / abs(float(want))
        print("[parity] e2e_rnn_ss %s hip %.6f ref %.6f rel %.2e" % (name, float(got), float(want), rel))
        assert rel < 1e-5
    assert abs(float(m.acc) - float(p["acc"])) < 1e-6
    loss.backward()
    check_grads(m, grads, tol=5e-4)


def test_e2e_rnn_vggblstm_golden():
    """BASELINE config 4's encoder type (etype vggblstm: VGG + stacked non-projected BLSTM + tanh(l_last),
    rnn/encoders.py:103-162) against the reference's own run: encoder states, losses, accuracy, every gradient"""
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn_vggblstm.npz"))
    m = load_sd(E2E(12, 7, _rnn_args(etype="vggblstm", elayers=2, eunits=12, eprojs=10)), sd)
    m.train()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    assert m.hlens == p["hlens"].tolist()
    report("e2e_rnn_vggblstm hs_pad", m.hs_pad, p["hs_pad"], 1e-4)
    for name, got, want in (("loss", loss, p["loss"]), ("loss_att", m.loss_att, p["loss_att"]), ("loss_ctc", m.loss_ctc, p["loss_ctc"])):
        rel = abs(float(got) - float(want)) / abs(float(want))
        print("[parity] e2e_rnn_vggblstm %s hip %.6f ref %.6f rel %.2e" % (name, float(got), float(want), rel))
        assert rel < 1e-5
    assert abs(float(m.acc) - float(p["acc"])) < 1e-6
    loss.backward()
    check_grads(m, grads, tol=5e-4)


def _config4_args(eunits, adim, dunits, aconv_chans=10, aconv_filts=100, elayers=3):
    return _rnn_args(elayers=elayers, subsample="1_1_1_1", etype="vggblstm", eunits=eunits, eprojs=eunits, dtype="lstm", dlayers=1,
                     dunits=dunits, atype="location", aheads=4, awin=5, aconv_chans=aconv_chans, aconv_filts=aconv_filts,
                     mtlalpha=0.5, adim=adim, char_list=None)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_config4_midsize_vs_oracle(prec):
    """BASELINE config 4's architecture (VGG-BLSTM x3, location-aware attention with the recipe's
    10 x 100 location filters, 1-layer LSTM decoder, CTC) at a width the one-launch LSTM step kernels and the fused attention
    kernels take (eunits = adim = dunits = 320), B = 4, T = 400, L = 12, V = 500 - one training step against the CPU oracle
    (e2e_asr.py:205-338 restated): encoder states, both losses, accuracy, the gradient of every parameter tensor"""
    import espnet_amd
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr import E2E
    from oracle import asr_oracle as oracle
    espnet_amd.set_precision(prec)
    try:
        torch.manual_seed(4)
        V = 500
        m = E2E(80, V, _config4_args(320, 320, 320))
        g = torch.Generator().manual_seed(44)
        B, T, L = 4, 400, 12
        ilens = [400, 371, 333, 260]
        xs = torch.randn(B, T, 80, generator=g)
        for i, n in enumerate(ilens):
            xs[i, n:] = 0.0
        ys = torch.randint(1, V - 1, (B, L), generator=g)
        ys[1, 9:] = -1
        ys[3, 7:] = -1
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        hs, hlens = oracle.rnn_encoder(sdr, "enc.", xs, ilens, 3, [1, 1, 1, 1], proj=False)
        loss_ctc = oracle.ctc_loss(oracle.linear(sdr, "ctc.ctc_lo.", hs), torch.tensor(hlens), ys)
        loss_att, acc, _ = oracle.rnn_att_decoder(sdr, "dec.", hs, hlens, ys, V - 1, V - 1, 1, "att.0.")
        want = 0.5 * loss_ctc + 0.5 * loss_att
        want.backward()
        m = m.to(DEV).train()
        flat = train.FlatParams(m)
        flat.expose_grads()
        flat.zero_grad()
        assert ops.lstm_step_ok(B, 320)
        loss = m(xs.to(DEV), ilens, ys.to(DEV))
        ops.wgrad_group_begin()
        try:
            loss.backward()
        finally:
            ops.wgrad_group_end()
        assert m.hlens == hlens
        ltol, gtol, htol = (1e-5, 2e-3, 1e-4) if prec == "fp32" else (2e-3, 6e-2, 2e-2)
        report("config 4 mid-size [%s] hs_pad" % prec, m.hs_pad, hs.detach(), htol)
        for name, got, ref in (("loss", loss, want), ("loss_att", m.loss_att, loss_att), ("loss_ctc", m.loss_ctc, loss_ctc)):
            rel = abs(float(got) - float(ref)) / abs(float(ref))
            print("[parity] config 4 mid-size [%s] %s hip %.6f oracle %.6f rel %.2e" % (prec, name, float(got), float(ref), rel))
            assert rel < ltol
        assert abs(float(m.acc) - acc) < (1e-6 if prec == "fp32" else 0.05)
        worst = ("", 0.0)
        gmax = max(float(v.grad.abs().max()) for v in sdr.values() if getattr(v, "grad", None) is not None)
        for name, prm in m.named_parameters():
            if sdr[name].grad is None:
                continue
            if float(sdr[name].grad.abs().max()) < 1e-6 * gmax:
                # mathematically zero (gvec.bias: the attention softmax is invariant to a shift of all energies): rounding
                # noise on both sides, compared by size
                assert float(prm.grad.abs().max()) < 1e-4 * gmax, name
                continue
            e = rel_err(prm.grad, sdr[name].grad)
            worst = max(worst, (name, e), key=lambda kv: kv[1])
            assert e < gtol, (name, e)
        print("[parity] config 4 mid-size [%s]: worst parameter-gradient rel err %.2e (%s)" % (prec, worst[1], worst[0]))
    finally:
        espnet_amd.set_precision("fp32")


def test_config4_fullsize_properties():
    """BASELINE config 4 at FULL size (VGG-BLSTM 3 x 1024, location attention 10 x 100, LSTM decoder 1024, B = 32, T = 1000,
    L = 100, V = 5000): properties the size does not change - a finite loss, the captured step's replay equals the eager step,
    two replays are bit-equal (no atomics race in loss or encoder states), every gradient finite"""
    import math
    import espnet_amd
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr import E2E
    espnet_amd.set_precision("fp32")
    torch.manual_seed(0)
    V = 5000
    m = E2E(80, V, _config4_args(1024, 1024, 1024)).to(DEV).train()
    flat = train.FlatParams(m)
    g = torch.Generator().manual_seed(0)
    B, T, L = 32, 1000, 100
    xs = torch.randn(B, T, 80, generator=g).to(DEV)
    ilens = [T - 7 * i for i in range(B)]
    ys = torch.randint(1, V - 1, (B, L), generator=g)

    def step():
        flat.zero_grad()
        loss = m(xs, ilens, ys)
        ops.wgrad_group_begin()
        try:
            loss.backward()
        finally:
            ops.wgrad_group_end()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = float(step())
        step()
        torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        loss = step()
    gr.replay()
    torch.cuda.synchronize()
    l1, hs1 = float(loss), m.hs_pad.clone()
    gr.replay()
    torch.cuda.synchronize()
    l2 = float(loss)
    print("[parity] config 4 full size: loss eager %.6f, replay %.6f / %.6f" % (eager, l1, l2))
    assert math.isfinite(eager) and abs(l1 - eager) <= 1e-5 * abs(eager)
    assert l1 == l2 and torch.equal(hs1, m.hs_pad)
    assert bool(torch.isfinite(flat.grad).all())


def test_e2e_rnn_golden_stacked_step_weight_gradients():
    """the same fixture with the gradients in a flat arena and backward inside wgrad_group_begin / end: the weight
    gradients of the per-step products (decoder LSTM cells, attention projections: M = batch rows) are stacked along
    the reduction and leave as one product per weight at the flush - same gradients as the reference"""
    from espnet_amd import ops, train
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn.npz"))
    m = load_sd(E2E(12, 7, _rnn_args()), sd).to(DEV)
    m.train()
    flat = train.FlatParams(m)
    flat.zero_grad()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    ops.wgrad_group_begin()
    try:
        loss.backward()
        assert len(ops._wgroup["stack"]) > 0, "no per-step weight gradient was stacked"
        nstacked = max(len(e[2]) for e in ops._wgroup["stack"].values())
    finally:
        ops.wgrad_group_end()
    assert nstacked > 1 and not ops._wgroup["stack"]
    flat.expose_grads()
    check_grads(m, grads, tol=5e-4)


@pytest.mark.parametrize("atype", ["dot", "add", "multi_head_dot", "multi_head_add", "multi_head_loc", "multi_head_multi_res_loc",
                                   "noatt", "coverage", "coverage_location", "location2d", "location_recurrent"])
def test_e2e_rnn_attention_types_golden(atype):
    """BLSTMP (frame subsampling 1_2) + additive / multi-head attentions == reference E2E on its own weights"""
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn_%s.npz" % atype))
    m = load_sd(E2E(9, 7, _rnn_args(etype="blstmp", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dlayers=1, dunits=10,
                                     atype=atype, adim=6, aheads=2, aconv_chans=3, aconv_filts=4)), sd)
    m.train()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    assert m.hlens == p["hlens"].tolist()
    report("e2e_rnn %s hs_pad" % atype, m.hs_pad, p["hs_pad"], 1e-4)
    for name, got, want in (("loss", loss, p["loss"]), ("loss_att", m.loss_att, p["loss_att"])):
        rel = abs(float(got) - float(want)) / abs(float(want))
        print("[parity] e2e_rnn %s %s hip %.6f ref %.6f rel %.2e" % (atype, name, float(got), float(want), rel))
        assert rel < 1e-5
    loss.backward()
    check_grads(m, grads, tol=5e-4)


def test_e2e_rnn_gru_golden():
    """bidirectional GRU-P encoder + 2-layer GRU attention decoder == reference E2E on its own weights"""
    from espnet_amd.nets.e2e_asr import E2E
    p, sd, grads = split_golden(load_golden("e2e_rnn_gru.npz"))
    m = load_sd(E2E(9, 7, _rnn_args(etype="bgrup", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dtype="gru", dlayers=2,
                                     dunits=10, atype="location", adim=6, aconv_chans=3, aconv_filts=4)), sd)
    m.train()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    assert m.hlens == p["hlens"].tolist()
    report("e2e_rnn gru hs_pad", m.hs_pad, p["hs_pad"], 1e-4)
    rel = abs(float(loss) - float(p["loss"])) / abs(float(p["loss"]))
    print("[parity] e2e_rnn gru loss hip %.6f ref %.6f rel %.2e" % (float(loss), float(p["loss"]), rel))
    assert rel < 1e-5
    loss.backward()
    check_grads(m, grads, tol=5e-4)


def _trn_args(**kw):
    d = dict(etype="vggblstmp", elayers=1, subsample="1_1", eunits=10, eprojs=8, dtype="lstm", dlayers=2, dunits=12,
             dec_embed_dim=6, dropout_rate=0.0, dropout_rate_decoder=0.0, dropout_rate_embed_decoder=0.0, joint_dim=7,
             joint_activation_type="tanh", rnnt_mode="rnnt", trans_type="warp-transducer", sym_space="<space>",
             sym_blank="<blank>", transformer_init="pytorch")
    d.update(kw)
    return argparse.Namespace(**d)


_TT = dict(dtype="transformer", dec_block_arch=[dict(type="transformer", d_hidden=16, d_ff=24, heads=2)], dec_block_repeat=2,
           transformer_dec_input_layer="embed", transformer_dec_pw_activation_type="relu")
_TRN_CASES = ["transducer_rnn.npz", "transducer_gru.npz", "transducer_conformer.npz", "transducer_tt.npz",
              "transducer_att.npz", "transducer_att_gru.npz"]


def _trn_case_args(name):
    if name == "transducer_rnn.npz":
        return _trn_args()
    if name == "transducer_gru.npz":
        return _trn_args(etype="bgru", elayers=2, dtype="gru", dlayers=2)
    if name == "transducer_att.npz":
        return _trn_args(rnnt_mode="rnnt-att", etype="blstmp", elayers=1, subsample="1_1", atype="location", adim=4, aheads=2,
                         awin=2, aconv_chans=2, aconv_filts=5)
    if name == "transducer_att_gru.npz":
        return _trn_args(rnnt_mode="rnnt-att", etype="blstmp", elayers=1, subsample="1_1", dtype="gru", dlayers=1,
                         atype="multi_head_add", adim=4, aheads=2, awin=2, aconv_chans=2, aconv_filts=5)
    if name == "transducer_tt.npz":
        return _trn_args(etype="transformer", enc_block_arch=[dict(type="transformer", d_hidden=64, d_ff=96, heads=4)],
                         enc_block_repeat=2, transformer_enc_input_layer="conv2d",
                         transformer_enc_self_attn_type="self_attn", transformer_enc_positional_encoding_type="abs_pos",
                         transformer_enc_pw_activation_type="relu", transformer_enc_conv_mod_activation_type="relu", **_TT)
    arch = [dict(type="conformer", d_hidden=64, d_ff=96, heads=4, macaron_style=True, use_conv_mod=True,
                 conv_mod_kernel=7)]
    return _trn_args(etype="transformer", enc_block_arch=arch, enc_block_repeat=2,
                     transformer_enc_input_layer="conv2d", transformer_enc_self_attn_type="rel_self_attn",
                     transformer_enc_positional_encoding_type="rel_pos", transformer_enc_pw_activation_type="swish",
                     transformer_enc_conv_mod_activation_type="swish", dlayers=1)


@pytest.mark.parametrize("name", _TRN_CASES)
def test_transducer_golden(name):
    """encoder -> DecoderRNNT -> JointNetwork == reference modules; loss == float64 transducer recursion"""
    from espnet_amd.nets.e2e_asr_transducer import E2E
    p, sd, grads = split_golden(load_golden(name))
    m = load_sd(E2E(12, 6, _trn_case_args(name)), sd)
    m.train()
    loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
    report(name + " hs_pad", m.hs_pad, p["hs_pad"], 1e-4)
    report(name + " joint logits", m.pred_pad, p["pred_pad"], 1e-4)
    rel = abs(float(loss) - float(p["loss"])) / abs(float(p["loss"]))
    print("[parity] %s loss hip %.6f ref %.6f rel %.2e" % (name, float(loss), float(p["loss"]), rel))
    assert rel < 1e-5
    loss.backward()
    check_grads(m, grads, tol=5e-4)


@pytest.mark.parametrize("chunk_rows", [1 << 15, 7])
@pytest.mark.parametrize("name", [n for n in _TRN_CASES if "att" not in n and "tt" not in n])
def test_transducer_fused_joint_loss_golden(name, chunk_rows):
    """JointRNNTLossFn (joint network + transducer loss streamed over lattice rows, the (B,T,U,V) logits never
    materialised) against the reference fixtures: same loss, same parameter gradients as the materialised path is held
    to; chunk_rows = 7 forces several frame chunks per utterance (accumulation of the prediction-side gradient)"""
    from espnet_amd.nets.e2e_asr_transducer import E2E
    from espnet_amd.nets.transducer.joint_network import JointNetwork
    p, sd, grads = split_golden(load_golden(name))
    m = load_sd(E2E(12, 6, _trn_case_args(name)), sd)
    m.train()
    m.fused_loss = True
    orig = JointNetwork.loss
    JointNetwork.loss = lambda self, *a, **k: orig(self, *a, **dict(k, chunk_rows=chunk_rows))
    try:
        loss = m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))
        assert m.pred_pad is None
        rel = abs(float(loss) - float(p["loss"])) / abs(float(p["loss"]))
        print("[parity] %s fused joint+loss (chunk_rows %d): hip %.6f ref %.6f rel %.2e" % (name, chunk_rows, float(loss), float(p["loss"]), rel))
        assert rel < 1e-5
        loss.backward()
        check_grads(m, grads, tol=5e-4)
    finally:
        JointNetwork.loss = orig


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_rnnt_loss_statistics_from_gemm_epilogue_match_stored_logits(prec):
    """JointRNNTLossFn with the node statistics taken from the logits GEMM's row-statistics epilogue (no logits written in
    the forward) against the same function storing the logits chunk: loss and every gradient"""
    import espnet_amd
    from espnet_amd import ops
    from espnet_amd import rnn_functional as R_
    espnet_amd.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(3)
        B, T, U, J, V = 3, 40, 9, 64, 777
        e0, d0 = torch.randn(B, T, J, generator=g).to(DEV), torch.randn(B, U, J, generator=g).to(DEV)
        w0, b0 = (torch.randn(V, J, generator=g) * 0.2).to(DEV), torch.randn(V, generator=g).to(DEV)
        labels = torch.randint(1, V, (B, U - 1), generator=g).to(torch.int32).to(DEV)
        tl, ul = [40, 33, 17], [8, 5, 8]
        tlens, ulens = torch.tensor(tl, dtype=torch.int32, device=DEV), torch.tensor(ul, dtype=torch.int32, device=DEV)
        res = {}
        for fused in (True, False):
            ops.RNNT_FUSED_STATS = fused
            e, d, w, b = (t.clone().requires_grad_(True) for t in (e0, d0, w0, b0))
            loss = R_.JointRNNTLossFn.apply(e, d, w, b, labels, tlens, ulens, 0, ops.ACT_TANH, tl, 150)
            loss.backward()
            res[fused] = (loss.detach().clone(), e.grad.clone(), d.grad.clone(), w.grad.clone(), b.grad.clone())
        tol = 2e-5 if prec == "fp32" else 2e-3
        for name, a_, b_ in zip(("loss", "d e", "d d", "d w_out", "d b_out"), res[True], res[False]):
            report("rnnt fused statistics %s %s" % (prec, name), a_, b_, tol)
    finally:
        ops.RNNT_FUSED_STATS = True
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_joint_loss_config5_shape_vs_materialised(prec):
    """the width of BASELINE config 5 (J = 320, V = 5000, U = 101) on a short ragged batch: JointRNNTLossFn against the
    materialised JointFn -> Linear -> RNNTLossFn path on the same operands: loss and all five gradients (fp32: 1e-5 /
    2e-4; bf16 operands: the two paths round identically up to the dZ cast, 2e-2)"""
    import espnet_amd
    from espnet_amd import rnn_functional as R
    from espnet_amd.nets.transducer.joint_network import JointNetwork
    espnet_amd.set_precision(prec)
    try:
        torch.manual_seed(5)
        B, T, U, De, Dd, J, V = 3, 41, 101, 256, 320, 320, 5000
        jn = JointNetwork(V, De, Dd, J, "tanh").to(DEV)
        g = torch.Generator().manual_seed(55)
        he, hd = torch.randn(B, T, De, generator=g).to(DEV), torch.randn(B, U, Dd, generator=g).to(DEV)
        y = torch.randint(1, V, (B, U - 1), generator=g).int().to(DEV)
        tl_host = [41, 33, 20]
        tl, ul = torch.tensor(tl_host, dtype=torch.int32).to(DEV), torch.tensor([100, 77, 51], dtype=torch.int32).to(DEV)
        res = []
        for fused in (True, False):
            jn.zero_grad()
            a, b = he.clone().requires_grad_(True), hd.clone().requires_grad_(True)
            if fused:
                loss = jn.loss(a, b, y, tl, ul, tl_host, 0, chunk_rows=1500)
            else:
                loss = R.RNNTLossFn.apply(jn(a, b).float(), y, tl, ul, 0)
            loss.backward()
            res.append((float(loss), a.grad, b.grad, [q.grad.clone() for q in jn.parameters()]))
        rel = abs(res[0][0] - res[1][0]) / abs(res[1][0])
        tol = 2e-4 if prec == "fp32" else 2e-2
        print("[parity] fused joint+loss [%s] J=320 V=5000: %.6f vs materialised %.6f (rel %.2e)" % (prec, res[0][0], res[1][0], rel))
        assert rel < (1e-5 if prec == "fp32" else 1e-4)
        report("fused joint+loss [%s] d h_enc" % prec, res[0][1], res[1][1], tol)
        report("fused joint+loss [%s] d h_dec" % prec, res[0][2], res[1][2], tol)
        for (n_, _q), ga, gb in zip(jn.named_parameters(), res[0][3], res[1][3]):
            report("fused joint+loss [%s] d %s" % (prec, n_), ga, gb, tol)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_joint_loss_config5_width_vs_oracle(prec):
    """the streamed joint network + transducer loss at BASELINE config 5's real width (J = 320, V = 5000,
    U = 101), GEMM row epilogues 7 / 8 on (the logits are never stored), against the ORACLE in float64: joint network
    lin_out(tanh(lin_enc(h_enc) + lin_dec(h_dec))) (transducer/joint_network.py) -> oracle.rnnt_loss (Graves 2012 recursion,
    brute-force-checked on small lattices; the loss VALUE stays 'parity unpinned by the package') -> autograd.
    Loss 1e-5 (fp32) / 1e-3 (bf16 operands), all five gradients."""
    import espnet_amd
    from espnet_amd.nets.transducer.joint_network import JointNetwork
    from oracle import asr_oracle as oracle
    espnet_amd.set_precision(prec)
    try:
        torch.manual_seed(5)
        B, T, U, De, Dd, J, V = 2, 41, 101, 256, 320, 320, 5000
        jn = JointNetwork(V, De, Dd, J, "tanh")
        g = torch.Generator().manual_seed(56)
        he, hd = 0.5 * torch.randn(B, T, De, generator=g), 0.5 * torch.randn(B, U, Dd, generator=g)
        y = torch.randint(1, V, (B, U - 1), generator=g).int()
        tl_host, ul_host = [41, 29], [100, 63]
        # ---- oracle, float64 ----
        sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in jn.state_dict().items()}
        a64, b64 = he.double().clone().requires_grad_(True), hd.double().clone().requires_grad_(True)
        pre = oracle.linear(sd, "lin_enc.", a64).unsqueeze(2) + torch.nn.functional.linear(b64, sd["lin_dec.weight"]).unsqueeze(1)
        z = oracle.linear(sd, "lin_out.", torch.tanh(pre))                                   # (B, T, U, V) float64: 0.33 GB
        want = oracle.rnnt_loss(z, y.long(), torch.tensor(tl_host), torch.tensor(ul_host), blank=0)
        want.backward()
        del z, pre
        # ---- HIP: streamed, row epilogues ----
        jn = jn.to(DEV)
        a, b = he.to(DEV).requires_grad_(True), hd.to(DEV).requires_grad_(True)
        tl, ul = torch.tensor(tl_host, dtype=torch.int32).to(DEV), torch.tensor(ul_host, dtype=torch.int32).to(DEV)
        loss = jn.loss(a, b, y.to(DEV), tl, ul, tl_host, 0, chunk_rows=1500)
        loss.backward()
        rel = abs(float(loss) - float(want)) / abs(float(want))
        print("[parity] fused joint+loss [%s] J=320 V=5000 vs float64 oracle: %.6f vs %.6f (rel %.2e)" % (prec, float(loss), float(want), rel))
        assert rel < (1e-5 if prec == "fp32" else 1e-3)
        tol = 2e-4 if prec == "fp32" else 3e-2
        report("fused joint+loss vs oracle [%s] d h_enc" % prec, a.grad, a64.grad, tol)
        report("fused joint+loss vs oracle [%s] d h_dec" % prec, b.grad, b64.grad, tol)
        for n_, q in jn.named_parameters():
            report("fused joint+loss vs oracle [%s] d %s" % (prec, n_), q.grad, sd[n_].grad, tol)
    finally:
        espnet_amd.set_precision("fp32")


@pytest.mark.parametrize("name", _TRN_CASES)
def test_transducer_decoding_golden(name):
    """greedy and default beam search (beam_search_transducer.py:130-237), with and without RNNLM fusion, reproduce the
    reference's recorded hypotheses on the trie / row-batched implementation:
    token sequences exactly, scores to 1e-4, in the reference's n-best order"""
    from espnet_amd.nets.beam_search_transducer import BeamSearchTransducer
    from espnet_amd.nets.e2e_asr_transducer import E2E
    p, sd, _ = split_golden(load_golden(name))
    args = _trn_case_args(name)
    m = load_sd(E2E(12, 6, args), sd)
    m.train()
    m(p["xs"].to(DEV), p["ilens"], p["ys"].to(DEV))      # as in the fixture: BatchNorm running stats see one training batch
    m.eval()
    x = p["xs"][0, : int(p["ilens"][0])].numpy()
    from espnet_amd.nets.lm import ClassifierWithState, RNNLM
    lm = ClassifierWithState(RNNLM(6, 1, 8, None, "lstm", 0.0))
    lm.load_state_dict({k[3:]: v for k, v in p.items() if k.startswith("lm/")})      # the reference LM's own keys
    lm.to(DEV).eval()
    for tag, kw in (("greedy", dict(beam_size=1)), ("beam3", dict(beam_size=3, search_type="default")),
                    ("beam3_nonorm", dict(beam_size=3, search_type="default", score_norm=False)),
                    ("beam3_lm", dict(beam_size=3, search_type="default", lm=lm, lm_weight=0.5, nstep=2))):
        if "dec_%s_lens" % tag not in p:
            continue
        nb = m.recognize(x, BeamSearchTransducer(decoder=m.decoder if hasattr(m, "decoder") else m.dec, **kw))
        nb = nb if isinstance(nb, list) else [nb]
        lens = p["dec_%s_lens" % tag].tolist()
        want_scores = p["dec_%s_scores" % tag].tolist()
        flat = p["dec_%s_yseq" % tag].tolist()
        assert len(nb) == len(lens), (tag, len(nb), len(lens))
        o = 0
        for h, n, sc in zip(nb, lens, want_scores):
            assert h["yseq"] == flat[o:o + n], (tag, h["yseq"], flat[o:o + n])      # token ids: bit exact
            assert abs(h["score"] - sc) <= 1e-4 * max(1.0, abs(sc)), (tag, h["score"], sc)
            o += n
        print("[parity] %s %s: %d hypotheses identical, best score %.5f (ref %.5f)" % (name, tag, len(nb), nb[0]["score"],
                                                                                         want_scores[0]))


_SMALL = dict(etype="blstmp", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dlayers=1, dunits=10, adim=6, aheads=2,
              aconv_chans=3, aconv_filts=4)


@pytest.mark.parametrize("name,idim,kw", [
    ("e2e_rnn.npz", 12, dict()),
    ("e2e_rnn_dot.npz", 9, dict(_SMALL, atype="dot")), ("e2e_rnn_add.npz", 9, dict(_SMALL, atype="add")),
    ("e2e_rnn_multi_head_dot.npz", 9, dict(_SMALL, atype="multi_head_dot")),
    ("e2e_rnn_multi_head_add.npz", 9, dict(_SMALL, atype="multi_head_add")),
    ("e2e_rnn_multi_head_loc.npz", 9, dict(_SMALL, atype="multi_head_loc")),
    ("e2e_rnn_multi_head_multi_res_loc.npz", 9, dict(_SMALL, atype="multi_head_multi_res_loc")),
    ("e2e_rnn_noatt.npz", 9, dict(_SMALL, atype="noatt")), ("e2e_rnn_coverage.npz", 9, dict(_SMALL, atype="coverage")),
    ("e2e_rnn_coverage_location.npz", 9, dict(_SMALL, atype="coverage_location")),
    ("e2e_rnn_location2d.npz", 9, dict(_SMALL, atype="location2d")),
    ("e2e_rnn_location_recurrent.npz", 9, dict(_SMALL, atype="location_recurrent")),
    ("e2e_rnn_gru.npz", 9, dict(etype="bgrup", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dtype="gru", dlayers=2,
                                dunits=10, atype="location", adim=6, aconv_chans=3, aconv_filts=4))])
def test_rnn_decoding_golden(name, idim, kw):
    """a20 decode: E2E.recognize -> Decoder.recognize_beam (rnn/decoders.py:313-605) - attention-only, joint CTC
    (prefix scores from the HIP kernel), LM fusion and CTC-only with a length cap - reproduces the reference's n-best:
    token sequences exactly, scores to 1e-4."""
    from espnet_amd.nets.e2e_asr import E2E
    from espnet_amd.nets.lm import ClassifierWithState, RNNLM
    p, sd, _ = split_golden(load_golden(name))
    m = load_sd(E2E(idim, 7, _rnn_args(**kw)), sd).eval()
    lm = ClassifierWithState(RNNLM(7, 1, 8, None, "lstm", 0.0))
    lm.load_state_dict({k[4:]: v for k, v in p.items() if k.startswith("rlm/")})
    lm.to(DEV).eval()
    x = p["xs"][0].numpy()
    for tag, rkw, use_lm in (("b3", dict(beam_size=3, ctc_weight=0.0, penalty=0.0), False),
                             ("b3ctc", dict(beam_size=3, ctc_weight=0.5, penalty=0.1), False),
                             ("b2lm", dict(beam_size=2, ctc_weight=0.3, penalty=0.0, lm_weight=0.4), True),
                             ("b3ctc1", dict(beam_size=3, ctc_weight=1.0, penalty=0.2, maxlenratio=0.5), False)):
        ra = argparse.Namespace(**dict(dict(nbest=3, maxlenratio=0.0, minlenratio=0.0, lm_weight=0.0), **rkw))
        nb = m.recognize(x, ra, _rnn_args().char_list, lm if use_lm else None)
        lens, flat, scores = p["rb_%s_lens" % tag].tolist(), p["rb_%s_yseq" % tag].tolist(), p["rb_%s_scores" % tag].tolist()
        assert len(nb) == len(lens), (tag, len(nb), len(lens))
        o = 0
        for h, n, sc in zip(nb, lens, scores):
            assert h["yseq"] == flat[o:o + n], (tag, h["yseq"], flat[o:o + n])
            assert abs(float(h["score"]) - sc) <= 1e-4 * max(1.0, abs(sc)), (tag, float(h["score"]), sc)
            o += n
        print("[parity] %s %s: %d hypotheses identical, best %.5f (ref %.5f)" % (name, tag, len(nb), float(nb[0]["score"]),
                                                                               scores[0]))


def test_rnn_batch_beam_search_golden():
    """a19 / a20: E2E.recognize_batch -> Decoder.recognize_beam_batch (rnn/decoders.py:632-974), the vectorised search over a
    batch of utterances x beam: attention only, joint CTC (CTCPrefixScoreTH arithmetic with the reference's CPU pre-selection
    of int(1.5 * beam) labels, which is what recorded the fixture), CTC + RNNLM fusion, a length-capped search with a minimum
    length - the reference's n-best per utterance: token sequences exactly, scores to 1e-4."""
    from espnet_amd.nets.e2e_asr import E2E
    from espnet_amd.nets.lm import ClassifierWithState, RNNLM
    p, sd, _ = split_golden(load_golden("e2e_rnn_batchbeam.npz"))
    m = load_sd(E2E(12, 7, _rnn_args()), sd).eval()
    lm = ClassifierWithState(RNNLM(7, 1, 8, None, "lstm", 0.0))
    lm.load_state_dict({k[4:]: v for k, v in p.items() if k.startswith("rlm/")})
    lm.to(DEV).eval()
    ilens = p["ilens"].tolist()
    feats = [p["xs"][b, : ilens[b]].numpy() for b in range(3)]
    for tag, rkw, use_lm in (("b3", dict(beam_size=3, ctc_weight=0.0, penalty=0.0), False),
                             ("b3ctc", dict(beam_size=3, ctc_weight=0.5, penalty=0.1), False),
                             ("b2lm", dict(beam_size=2, ctc_weight=0.3, penalty=0.0, lm_weight=0.4), True),
                             ("b4len", dict(beam_size=4, ctc_weight=0.3, penalty=0.2, maxlenratio=0.4, minlenratio=0.1), False)):
        ra = argparse.Namespace(**dict(dict(nbest=2, maxlenratio=0.0, minlenratio=0.0, lm_weight=0.0, ctc_window_margin=0), **rkw))
        nb = m.recognize_batch(feats, ra, _rnn_args().char_list, lm if use_lm else None,
                               ctc_scoring_num=int(ra.beam_size * 1.5))
        counts, lens = p["bb_%s_n" % tag].tolist(), p["bb_%s_lens" % tag].tolist()
        flat, scores = p["bb_%s_yseq" % tag].tolist(), p["bb_%s_scores" % tag].tolist()
        assert [len(u) for u in nb] == counts, (tag, [len(u) for u in nb], counts)
        o = q = 0
        for u in nb:
            for hyp in u:
                n, sc = lens[q], scores[q]
                assert hyp["yseq"] == flat[o:o + n], (tag, q, hyp["yseq"], flat[o:o + n])
                assert abs(float(hyp["score"]) - sc) <= 1e-4 * max(1.0, abs(sc)), (tag, q, float(hyp["score"]), sc)
                o += n
                q += 1
        print("[parity] recognize_beam_batch %s: %d hypotheses of 3 utterances identical, best %s (ref %s)"
              % (tag, q, [round(float(u[0]["score"]), 4) for u in nb], [round(scores[i], 4) for i in (0, counts[0], counts[0] + counts[1])]))
    # the reference's rule for device tensors (CTC scores for the whole vocabulary) runs too and agrees on the best hypothesis
    ra = argparse.Namespace(nbest=1, maxlenratio=0.0, minlenratio=0.0, lm_weight=0.0, ctc_window_margin=0, beam_size=3,
                            ctc_weight=0.5, penalty=0.1)
    full = m.recognize_batch(feats, ra, _rnn_args().char_list, None)
    pre = m.recognize_batch(feats, ra, _rnn_args().char_list, None, ctc_scoring_num=4)
    assert [u[0]["yseq"] for u in full] == [u[0]["yseq"] for u in pre]


@pytest.mark.parametrize("tag,cls,kw", [
    ("rnnp", "RNNEncoder", dict(num_layers=3, hidden_size=12, output_size=10, subsample=(2, 1))),
    ("gru", "RNNEncoder", dict(rnn_type="gru", bidirectional=False, use_projection=False, num_layers=2, hidden_size=12,
                               output_size=10, subsample=None)),
    ("vgg", "VGGRNNEncoder", dict(num_layers=1, hidden_size=12, output_size=10))])
def test_espnet2_rnn_encoders_golden(tag, cls, kw):
    """espnet2 RNNEncoder / VGGRNNEncoder (the espnet1 stacks behind AbsEncoder): outputs, lengths and all parameter
    gradients against the reference's own encoders"""
    import espnet_amd.espnet2 as e2
    p, sd, grads = split_golden(load_golden("enc2_%s.npz" % tag))
    enc = getattr(e2, cls)(20, **kw)
    assert list(enc.state_dict().keys()) == list(sd.keys())
    enc = load_sd(enc, sd)
    enc.train()
    y, olens, _ = enc(p["xs"].to(DEV), p["ilens"])
    assert olens.tolist() == p["olens"].tolist() and enc.output_size() == 10
    report("espnet2 %s fwd" % tag, y, p["y"], 2e-5)
    y.backward(p["gy"].to(DEV))
    check_grads(enc, grads, tol=5e-4)


@pytest.mark.parametrize("tag,dkw", [
    ("loc", dict(rnn_type="lstm", num_layers=2, att_conf=dict(atype="location", adim=8, aconv_chans=3, aconv_filts=4))),
    ("mh", dict(rnn_type="gru", num_layers=1, context_residual=True, att_conf=dict(atype="multi_head_add", adim=8, aheads=2)))])
def test_espnet2_rnn_model_golden(tag, dkw):
    """ESPnetASRModel(RNNEncoder, attention RNNDecoder, CTC): loss, stats, all parameter gradients and the BeamSearch n-best
    (the decoder is a plain ScorerInterface: scored hypothesis by hypothesis) against the reference's own espnet2 model"""
    from espnet_amd.espnet2 import CTC, ESPnetASRModel, RNNDecoder, RNNEncoder
    from espnet_amd.nets.beam_search import BeamSearch
    from espnet_amd.nets.ctc_prefix_score import CTCPrefixScorer, LengthBonus
    p, sd, grads = split_golden(load_golden("espnet2_rnn_%s.npz" % tag))
    model = ESPnetASRModel(vocab_size=30, encoder=RNNEncoder(20, num_layers=2, hidden_size=12, output_size=10, subsample=(2, 1)),
                           decoder=RNNDecoder(30, 10, hidden_size=12, **dkw), ctc=CTC(30, 10, ctc_type="builtin"),
                           ctc_weight=0.3, lsm_weight=0.1)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model = load_sd(model, sd)
    model.train()
    loss, stats, weight = model(p["speech"].to(DEV), p["speech_lengths"], p["text"].to(DEV), p["text_lengths"])
    for k in ("loss", "loss_att", "loss_ctc"):
        got, want = float(stats[k]), float(p[k])
        print("[parity] espnet2 rnn %s %s hip %.6f ref %.6f" % (tag, k, got, want))
        assert abs(got - want) <= 1e-5 * abs(want)
    assert abs(float(stats["acc"]) - float(p["acc"])) < 1e-6
    loss.backward()
    check_grads(model, grads, tol=5e-4)
    model.eval()
    with torch.no_grad():
        enc, _ = model.encode(p["speech"][:1].to(DEV), p["speech_lengths"][:1])
    for btag, cw in (("w00", 0.0), ("w03", 0.3)):
        scorers = dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos), length_bonus=LengthBonus(30))
        bs = BeamSearch(scorers, dict(decoder=1.0 - cw, ctc=cw, length_bonus=0.1), 3, 30, model.sos, model.eos,
                        pre_beam_score_key="full")
        got = bs(enc[0])[:3]
        lens, flat, scores = p["beam_%s_lens" % btag].tolist(), p["beam_%s_yseq" % btag].tolist(), p["beam_%s_scores" % btag].tolist()
        want, o = [], 0
        for n in lens:
            want.append(flat[o:o + n])
            o += n
        print("[parity] espnet2 rnn %s beam %s: hip %s ref %s" % (tag, btag, [round(float(h.score), 4) for h in got],
                                                                 [round(s, 4) for s in scores]))
        assert [h.yseq.tolist() for h in got] == want
        for h, s in zip(got, scores):
            assert abs(float(h.score) - s) <= 1e-4 * max(1.0, abs(s))


@pytest.mark.parametrize("shape", [(32, 1024, 9, True), (16, 320, 7, False), (5, 64, 11, True), (48, 128, 5, True)])
def test_lstm_fused_step_kernels(shape):
    """the one-launch LSTM step kernels (eamd_lstm_step_fwd / _bwd: fp32-MFMA recurrent product + cell in one launch)
    inside LSTMSeqFn against torch.nn.LSTM in float64 (outputs, d input projections, d W_hh, d b_hh), forward and reversed
    direction, with and without packed-sequence masking, and against the GEMM + cell-kernel steps they replace"""
    from espnet_amd import ops
    from espnet_amd import rnn_functional as R
    B, H, T, masked = shape
    g = torch.Generator().manual_seed(B + H)
    ref = torch.nn.LSTM(H, H, 1).double()
    w_hh, b_hh = ref.weight_hh_l0.detach().float().to(DEV), ref.bias_hh_l0.detach().float().to(DEV)
    x = torch.randn(T, B, H, generator=g).double()
    lens = sorted([max(1, T - (i * T) // (2 * B)) for i in range(B)], reverse=True) if masked else [T] * B
    live = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).to(torch.uint8).to(DEV).contiguous() if masked else None
    gy = torch.randn(T, B, H, generator=g)
    for t in range(T):
        for b in range(B):
            if t >= lens[b]:
                gy[t, b] = 0.0
    # float64 reference through pack_padded_sequence (what `live` reproduces)
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, torch.tensor(lens), enforce_sorted=True)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(ref(packed)[0], total_length=T)
    yr.backward(gy.double())
    gx_ref = (xr.detach() @ ref.weight_ih_l0.detach().t() + ref.bias_ih_l0.detach()).float().to(DEV)
    res = {}
    for fused in (True, False):
        ops.LSTM_FUSED_STEP = fused
        try:
            gx = gx_ref.clone().requires_grad_(True)
            wh, bh = w_hh.clone().requires_grad_(True), b_hh.clone().requires_grad_(True)
            y = R.LSTMSeqFn.apply(gx, wh, bh, live, False)
            y.backward(gy.to(DEV))
            res[fused] = (y.detach(), gx.grad, wh.grad, bh.grad)
        finally:
            ops.LSTM_FUSED_STEP = True
    assert ops.lstm_step_ok(B, H)
    report("lstm fused step y %s" % (shape,), res[True][0], yr.detach(), 2e-6)
    report("lstm fused step dW_hh %s" % (shape,), res[True][2], ref.weight_hh_l0.grad, 2e-5)
    report("lstm fused step db_hh %s" % (shape,), res[True][3], ref.bias_hh_l0.grad, 2e-5)
    # d gx = d gates: against the unfused HIP path and, through W_ih, against the float64 input gradient
    report("lstm fused step dgates vs GEMM + cell path %s" % (shape,), res[True][1], res[False][1], 2e-5)
    report("lstm fused step dx %s" % (shape,), res[True][1].double().cpu() @ ref.weight_ih_l0.detach(), xr.grad, 2e-5)
    # reversed direction: equals the forward direction on the time-reversed sequence when nothing is masked
    if not masked:
        gx = gx_ref.clone()
        yb = R.LSTMSeqFn.apply(gx, w_hh, b_hh, None, True)
        yf = R.LSTMSeqFn.apply(gx.flip(0).contiguous(), w_hh, b_hh, None, False)
        report("lstm fused step reverse %s" % (shape,), yb, yf.flip(0), 1e-6)


@pytest.mark.parametrize("shape", [(32, 1024, 12, True, 2), (32, 1024, 6, False, 1), (16, 320, 7, True, 2), (5, 64, 11, True, 2),
                                   (48, 128, 5, True, 1), (64, 256, 9, True, 2), (20, 512, 40, True, 2)])
def test_lstm_persistent_sequence_kernels(shape):
    """eamd_lstm_seq_fwd / _bwd (csrc/lstm_seq.hip: all time steps of one or two recurrences in ONE persistent launch,
    recurrent weights in registers, h_t / dgates_t handed between workgroups through write-through stores + flags)
    against torch.nn.LSTM(bidirectional) in float64 on packed sequences, and against the per-step kernels; every
    hand-off wait completed (status word 0), and a second launch reproduces the first bit for bit (nothing stale is
    ever read)."""
    from espnet_amd import ops
    from espnet_amd import rnn_functional as R
    B, H, T, masked, ndir = shape
    g = torch.Generator().manual_seed(B + H + T)
    ref = torch.nn.LSTM(H, H, 1, bidirectional=(ndir == 2)).double()
    x = torch.randn(T, B, H, generator=g).double()
    lens = sorted([max(1, T - (i * T) // (2 * B)) for i in range(B)], reverse=True) if masked else [T] * B
    live = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).to(torch.uint8).to(DEV).contiguous() if masked else None
    gy = torch.randn(T, B, ndir * H, generator=g)
    for t in range(T):
        for b in range(B):
            if t >= lens[b]:
                gy[t, b] = 0.0
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, torch.tensor(lens), enforce_sorted=True)
    yr, _ = torch.nn.utils.rnn.pad_packed_sequence(ref(packed)[0], total_length=T)
    yr.backward(gy.double())
    sfx = ["", "_reverse"][:ndir]
    w = {s: (getattr(ref, "weight_hh_l0" + s).detach().float().to(DEV), getattr(ref, "bias_hh_l0" + s).detach().float().to(DEV))
         for s in sfx}
    gx_ref = {s: (x @ getattr(ref, "weight_ih_l0" + s).detach().t() + getattr(ref, "bias_ih_l0" + s).detach()).float().to(DEV)
              for s in sfx}
    assert ops.lstm_seq_ok(ndir, B, H)

    def run(persistent):
        ops.LSTM_PERSISTENT = persistent
        try:
            leaves = []
            if persistent:
                flat = []
                for i, s in enumerate(sfx):
                    gx, wh, bh = gx_ref[s].clone().requires_grad_(True), w[s][0].clone().requires_grad_(True), w[s][1].clone().requires_grad_(True)
                    leaves.append((gx, wh, bh))
                    flat += [gx, wh, bh, i == 1]
                ys = R.LSTMSeqGroupFn.apply(live, ndir, *flat)
                st_f = ops.lstm_seq_status()
                torch.autograd.backward(list(ys), [gy[..., i * H:(i + 1) * H].contiguous().to(DEV) for i in range(ndir)])
                st_b = ops.lstm_seq_status()
                assert st_f == 0 and st_b == 0, (st_f, st_b)
            else:
                ys = []
                for i, s in enumerate(sfx):
                    gx, wh, bh = gx_ref[s].clone().requires_grad_(True), w[s][0].clone().requires_grad_(True), w[s][1].clone().requires_grad_(True)
                    leaves.append((gx, wh, bh))
                    ys.append(R.LSTMSeqFn.apply(gx, wh, bh, live, i == 1))
                torch.autograd.backward(ys, [gy[..., i * H:(i + 1) * H].contiguous().to(DEV) for i in range(ndir)])
            return [y.detach() for y in ys], leaves
        finally:
            ops.LSTM_PERSISTENT = True

    ys_p, lv_p = run(True)
    ys_p2, lv_p2 = run(True)
    ys_s, lv_s = run(False)
    for i, s in enumerate(sfx):
        tag = "%s dir %d" % (shape, i)
        report("lstm persistent y " + tag, ys_p[i], yr.detach()[..., i * H:(i + 1) * H], 2e-6)
        report("lstm persistent dW_hh " + tag, lv_p[i][1].grad, getattr(ref, "weight_hh_l0" + s).grad, 2e-5)
        report("lstm persistent db_hh " + tag, lv_p[i][2].grad, getattr(ref, "bias_hh_l0" + s).grad, 2e-5)
        report("lstm persistent dgates vs step kernels " + tag, lv_p[i][0].grad, lv_s[i][0].grad, 2e-5)
        assert torch.equal(ys_p[i], ys_p2[i]) and torch.equal(lv_p[i][0].grad, lv_p2[i][0].grad), "persistent launch not reproducible " + tag
        report("lstm persistent y vs step kernels " + tag, ys_p[i], ys_s[i], 1e-6)
    dx = sum(lv_p[i][0].grad.double().cpu() @ getattr(ref, "weight_ih_l0" + s).detach() for i, s in enumerate(sfx))
    report("lstm persistent dx %s" % (shape,), dx, xr.grad, 2e-5)


def test_config5_fullsize_properties():
    """BASELINE config 5 at FULL size (RNN-Transducer: 12-block Conformer encoder d 256, 1-layer LSTM predictor 512, joint 320,
    B = 16, T = 1500 -> T' = 374, U = 101, V = 5000; 12.1 GB of joint logits if they were materialised): properties the size does
    not change - a finite loss, the captured step's replay equals the eager step, two replays bit-equal in the loss and the
    encoder states, every gradient finite - and the streamed joint / loss keeps the peak memory of the step below 32 GB
    (the reference materialises the (B, T', U, V) logits: transducer/rnn_decoder.py:160-165)."""
    import argparse
    import math
    import espnet_amd
    from espnet_amd import graphs, ops, train
    from espnet_amd.nets.e2e_asr_transducer import E2E
    espnet_amd.set_precision("fp32")
    arch = [dict(type="conformer", d_hidden=256, d_ff=2048, heads=4, macaron_style=True, use_conv_mod=True, conv_mod_kernel=31)]
    ns = argparse.Namespace(etype="transformer", enc_block_arch=arch, enc_block_repeat=12, transformer_enc_input_layer="conv2d",
                            transformer_enc_self_attn_type="rel_self_attn", transformer_enc_positional_encoding_type="rel_pos",
                            transformer_enc_pw_activation_type="swish", transformer_enc_conv_mod_activation_type="swish",
                            dtype="lstm", dlayers=1, dunits=512, dec_embed_dim=512, joint_dim=320, joint_activation_type="tanh",
                            dropout_rate_decoder=0.0, dropout_rate_embed_decoder=0.0, rnnt_mode="rnnt", trans_type="warp-transducer",
                            sym_space="<space>", sym_blank="<blank>", transformer_init="pytorch")
    torch.manual_seed(0)
    V, B, T, L = 5000, 16, 1500, 100
    m = E2E(80, V, ns).to(DEV).train()
    flat = train.FlatParams(m)
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(B, T, 80, generator=g).to(DEV)
    ilens = [T - 11 * i for i in range(B)]
    ys = torch.randint(1, V - 1, (B, L), generator=g)        # labels are parsed on the host (as in the reference): no device sync
    torch.cuda.reset_peak_memory_stats()

    def step():
        flat.zero_grad()
        loss = m(xs, ilens, ys)
        ops.wgrad_group_begin()
        try:
            loss.backward()
        finally:
            ops.wgrad_group_end()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = float(step())
        step()
        torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    gr = graphs.new_graph()
    with torch.cuda.graph(gr):
        loss = step()
    kinds = graphs.audit(gr, "config 5 step graph")
    gr.replay()
    torch.cuda.synchronize()
    l1, g1 = float(loss), flat.grad.clone()
    gr.replay()
    torch.cuda.synchronize()
    l2 = float(loss)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print("[parity] config 5 full size: loss eager %.6f, replay %.6f / %.6f; peak memory %.1f GB; graph nodes %s" % (eager, l1, l2, peak, kinds))
    assert math.isfinite(eager) and abs(l1 - eager) <= 1e-5 * abs(eager)
    assert l1 == l2
    assert bool(torch.isfinite(flat.grad).all())
    rel = float((flat.grad.double() - g1.double()).norm() / g1.double().norm())
    assert rel < 1e-5, rel            # (split-K / column-sum atomics reorder fp32 sums between replays: not bit-equal, equal to rounding)
    assert peak < 32.0, peak
