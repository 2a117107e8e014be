"""Pins oracle/asr_oracle.py against vectors recorded from the reference (oracle/gen_golden.py)."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, split_golden

TOL = dict(rtol=2e-5, atol=2e-6)


def close(a, b, **kw):
    t = dict(TOL)
    t.update(kw)
    torch.testing.assert_close(a, b, **t)


def run_grads(y, gy, sd):
    params = [v for v in sd.values() if v.requires_grad]
    y.backward(gy)


def req(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}


def check_param_grads(sd, grads, **kw):
    for k, g in grads.items():
        assert sd[k].grad is not None, k
        close(sd[k].grad, g, **kw)


def test_layernorm(oracle):
    p, sd, grads = split_golden(load_golden("layernorm.npz"))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.layer_norm(sd, "", x)
    close(y, p["y"])
    y.backward(p["gy"])
    close(x.grad, p["gx"])
    check_param_grads(sd, grads)


def test_rel_shift(oracle):
    p, _, _ = split_golden(load_golden("rel_shift.npz"))
    assert torch.equal(oracle.rel_shift(p["x"]), p["y"])
    assert torch.equal(oracle.rel_shift(p["x2"]), p["y2"])
    # index formula of SURVEY.md §7: out[i,j] = P_flat[T1 + i*T2 + j], P = [0 | x]
    x = p["x"][0, 0]
    t1, t2 = x.shape
    for i in range(t1):
        for j in range(t2):
            f = t1 + i * t2 + j
            r, c = divmod(f, t2 + 1)
            want = 0.0 if c == 0 else float(x[r, c - 1])
            assert float(p["y"][0, 0, i, j]) == want


def test_rel_mha(oracle):
    p, sd, grads = split_golden(load_golden("rel_mha.npz"))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.rel_mha(sd, "", x, p["pos"], p["mask"], 4)
    close(y, p["y"])
    y.backward(p["gy"])
    close(x.grad, p["gx"])
    check_param_grads(sd, grads)


def test_mha(oracle):
    p, sd, grads = split_golden(load_golden("mha.npz"))
    sd = req(sd)
    q = p["q"].clone().requires_grad_(True)
    mem = p["mem"].clone().requires_grad_(True)
    y = oracle.mha(sd, "", q, mem, mem, p["mmask"], 4)
    close(y, p["y"])
    y.backward(p["gy"])
    close(q.grad, p["gq"])
    close(mem.grad, p["gmem"])
    check_param_grads(sd, grads)
    y2 = oracle.mha(sd, "", p["q"], p["q"], p["q"], p["cmask"], 4)
    close(y2, p["y_self"])


@pytest.mark.parametrize("name", ["swish", "relu"])
def test_ffn(oracle, name):
    p, sd, grads = split_golden(load_golden("ffn_%s.npz" % name))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.ffn(sd, "", x, oracle.activation(name))
    close(y, p["y"])
    y.backward(p["gy"])
    close(x.grad, p["gx"])
    check_param_grads(sd, grads)


def test_conv_module(oracle):
    p, sd, grads = split_golden(load_golden("conv_module.npz"))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    st = {}
    y = oracle.conv_module(sd, "", x, oracle.swish, True, st)
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    y.backward(p["gy"])
    close(x.grad, p["gx"], rtol=1e-4, atol=1e-5)
    check_param_grads(sd, grads, rtol=1e-4, atol=1e-5)
    close(st["norm.running_mean"], p["sd_after/norm.running_mean"])
    close(st["norm.running_var"], p["sd_after/norm.running_var"])
    sd_eval = {k: v.detach() for k, v in sd.items()}
    sd_eval["norm.running_mean"] = p["sd_after/norm.running_mean"]
    sd_eval["norm.running_var"] = p["sd_after/norm.running_var"]
    close(oracle.conv_module(sd_eval, "", p["x"], oracle.swish, False), p["y_eval"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["abs", "rel"])
def test_subsampling(oracle, name):
    p, sd, grads = split_golden(load_golden("subsampling_%s.npz" % name))
    sd = req(sd)
    y, pos, m = oracle.conv2d_subsampling(sd, "", p["x"], p["mask"], name == "rel")
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    assert torch.equal(m, p["ymask"])
    if name == "rel":
        close(pos, p["pos"])
    y.backward(p["gy"])
    check_param_grads(sd, grads, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("macaron", [0, 1])
@pytest.mark.parametrize("cnn", [0, 1])
def test_conformer_layer(oracle, macaron, cnn):
    p, sd, grads = split_golden(load_golden("conformer_layer_m%d_c%d.npz" % (macaron, cnn)))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.conformer_layer(sd, "", x, p["pos"], p["mask"], dict(aheads=4), True)
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    y.backward(p["gy"])
    close(x.grad, p["gx"], rtol=1e-4, atol=1e-5)
    check_param_grads(sd, grads, rtol=1e-4, atol=1e-5)


def test_lsm_loss(oracle):
    p, _, _ = split_golden(load_golden("lsm_loss.npz"))
    for norm in (0, 1):
        x = p["x"].clone().requires_grad_(True)
        loss = oracle.label_smoothing_loss(x, p["t"], 0.1, -1, bool(norm))
        close(loss, p["loss_n%d" % norm])
        loss.backward()
        close(x.grad, p["gx_n%d" % norm])
    close(oracle.label_smoothing_loss(p["x"], p["t"], 0.0, -1, False), p["loss_s0"])
    assert abs(oracle.accuracy(p["x"].view(-1, 17), p["t"], -1) - float(p["acc"])) < 1e-7


def test_ctc(oracle):
    p, sd, grads = split_golden(load_golden("ctc.npz"))
    for builtin in (True, False):
        lg = p["logits"].clone().requires_grad_(True)
        loss = oracle.ctc_loss(lg, p["hlens"], p["ys"], -1, builtin)
        close(loss, p["loss"])
        close(loss, p["loss_direct"])
        loss.backward()
        close(lg.grad, p["glogits"], rtol=1e-4, atol=1e-6)
    # per-utterance recursion and the infeasible case
    logp = torch.log_softmax(p["logits"], 2)
    ys = [[1, 2], [3, 3, 4], [2]]
    for b in range(3):
        close(oracle.ctc_alpha_nll(logp[b, : int(p["hlens"][b])], ys[b]), p["nll"][b])
    assert math.isinf(float(p["loss_infeasible"]))
    assert math.isinf(float(oracle.ctc_alpha_nll(logp[0, :3], [3, 3, 4])))
    # whole module: ctc_lo + loss
    sdr = req(sd)
    hs = p["hs"].clone().requires_grad_(True)
    loss = oracle.ctc_loss(oracle.linear(sdr, "ctc_lo.", hs), p["hlens"], p["ys"])
    loss.backward()
    close(hs.grad, p["ghs"], rtol=1e-4, atol=1e-6)
    check_param_grads(sdr, grads, rtol=1e-4, atol=1e-6)
    assert torch.equal(oracle.linear(sd, "ctc_lo.", p["hs"]).argmax(-1), p["argmax"])


def test_decoder(oracle):
    p, sd, grads = split_golden(load_golden("decoder.npz"))
    sd = req(sd)
    mem = p["mem"].clone().requires_grad_(True)
    u = p["ys_in"].shape[1]
    tmask = torch.tril(torch.ones(u, u, dtype=torch.bool)).unsqueeze(0)
    y = oracle.decoder(sd, "", p["ys_in"], tmask, mem, p["mmask"], 4)
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    y.backward(p["gy"])
    close(mem.grad, p["gmem"], rtol=1e-4, atol=1e-5)
    check_param_grads(sd, grads, rtol=1e-4, atol=2e-5)
    # cached one-step == last row of the full causal forward (test_transformer_decode.py:13-79)
    sdd = {k: v.detach() for k, v in sd.items()}
    for i in range(1, 5):
        tm = torch.tril(torch.ones(i, i, dtype=torch.bool)).unsqueeze(0)
        lp = torch.log_softmax(oracle.decoder(sdd, "", p["ys_in"][:1, :i], tm, p["mem"][:1], None, 4)[:, -1], -1)
        close(lp, p["step_logp"][i - 1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,cfg", [
    ("e2e_conformer.npz", dict(conformer=True, rel_pos=True, activation="swish")),
    ("e2e_transformer.npz", dict(conformer=False, rel_pos=False)),
])
def test_e2e(oracle, name, cfg):
    p, sd, grads = split_golden(load_golden(name))
    sd = req(sd)
    cfg = dict(cfg, aheads=4, mtlalpha=0.3, lsm_weight=0.1, odim=50)
    out = oracle.e2e_forward(sd, p["xs"], p["ilens"].tolist(), p["ys"], cfg, training=True)
    assert abs(float(out["loss"].detach()) - float(p["loss"])) <= 2e-5 * abs(float(p["loss"]))
    assert abs(out["acc"] - float(p["acc"])) < 1e-6
    close(out["hs_pad"], p["hs_pad"], rtol=1e-3, atol=1e-4)
    out["loss"].backward()
    worst = 0.0
    for k, g in grads.items():
        a = sd[k].grad
        assert a is not None, k
        if float(g.norm()) > 1e-3:   # e.g. linear_k.bias has an analytically zero gradient: cosine of noise
            cos = float(torch.nn.functional.cosine_similarity(a.flatten().double(), g.flatten().double(), dim=0))
            worst = max(worst, 1 - cos)
        close(a, g, rtol=2e-3, atol=2e-4)
    assert worst < 1e-5
    if "greedy" in p:
        hs, _ = oracle.encoder({k: v.detach() for k, v in sd.items()}, "encoder.", p["xs"][:1], None, cfg, training=False)
        ids = oracle.greedy_ctc(oracle.linear(sd, "ctc.ctc_lo.", hs)[0])
        assert ids == p["greedy"].tolist()


# ---- RNN paths (rows a20 / a21) -------------------------------------------------------------------
def _grad_check(sd, grads, rtol=2e-3, atol=2e-5):
    for k, g in grads.items():
        a = sd[k].grad
        if a is None:
            assert float(g.abs().max()) == 0.0, k
            continue
        close(a, g, rtol=rtol, atol=atol)


def test_e2e_rnn(oracle):
    """VGG-BLSTMP encoder + AttLoc LSTM decoder + CTC against the reference E2E (e2e_asr.py:205-338)"""
    p, sd, grads = split_golden(load_golden("e2e_rnn.npz"))
    sd = req(sd)
    hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 1, 1])   # vgg*: no RNNP subsampling (nets_utils.py:415-424)
    assert hlens == p["hlens"].tolist()
    close(hs, p["hs_pad"], rtol=1e-4, atol=1e-5)
    loss_ctc = oracle.ctc_loss(oracle.linear(sd, "ctc.ctc_lo.", hs), torch.tensor(hlens), p["ys"])
    loss_att, acc, _ = oracle.rnn_att_decoder(sd, "dec.", hs, hlens, p["ys"], 6, 6, 2, "att.0.")
    close(loss_ctc.detach(), p["loss_ctc"], rtol=1e-4, atol=1e-5)
    close(loss_att.detach(), p["loss_att"], rtol=1e-4, atol=1e-5)
    assert abs(acc - float(p["acc"])) < 1e-6
    loss = 0.5 * loss_ctc + 0.5 * loss_att
    close(loss.detach(), p["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    _grad_check(sd, grads)


@pytest.mark.parametrize("atype", ["dot", "add", "multi_head_dot", "multi_head_add", "multi_head_loc", "multi_head_multi_res_loc",
                                   "noatt", "coverage", "coverage_location", "location2d", "location_recurrent"])
def test_e2e_rnn_attention_types(oracle, atype):
    """BLSTMP (subsample 1_2) + the other attention types on the HIP path against the reference E2E"""
    p, sd, grads = split_golden(load_golden("e2e_rnn_%s.npz" % atype))
    sd = req(sd)
    hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 2, 1], vgg=False)
    assert hlens == p["hlens"].tolist()
    close(hs, p["hs_pad"], rtol=1e-4, atol=1e-5)
    loss_ctc = oracle.ctc_loss(oracle.linear(sd, "ctc.ctc_lo.", hs), torch.tensor(hlens), p["ys"])
    loss_att, acc, _ = oracle.rnn_att_decoder(sd, "dec.", hs, hlens, p["ys"], 6, 6, 1, "att.0.", atype=atype, aheads=2)
    close(loss_att.detach(), p["loss_att"], rtol=1e-4, atol=1e-5)
    loss = 0.5 * loss_ctc + 0.5 * loss_att
    close(loss.detach(), p["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    _grad_check(sd, grads)


def test_e2e_rnn_gru(oracle):
    """bidirectional GRU-P encoder + 2-layer GRU attention decoder against the reference E2E"""
    p, sd, grads = split_golden(load_golden("e2e_rnn_gru.npz"))
    sd = req(sd)
    hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 2, 1], vgg=False, gru=True)
    assert hlens == p["hlens"].tolist()
    close(hs, p["hs_pad"], rtol=1e-4, atol=1e-5)
    loss_ctc = oracle.ctc_loss(oracle.linear(sd, "ctc.ctc_lo.", hs), torch.tensor(hlens), p["ys"])
    loss_att, acc, _ = oracle.rnn_att_decoder(sd, "dec.", hs, hlens, p["ys"], 6, 6, 2, "att.0.", gru=True)
    loss = 0.5 * loss_ctc + 0.5 * loss_att
    close(loss.detach(), p["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    _grad_check(sd, grads)


def test_rnnt_loss_brute_force(oracle):
    """the lattice recursion equals the explicit sum over all alignments (Graves 2012, eq. 1-3)"""
    g = torch.Generator().manual_seed(7)
    for T, U, V in ((1, 1, 3), (3, 1, 4), (1, 3, 4), (4, 3, 5), (3, 4, 6)):
        z = torch.randn(1, T, U, V, generator=g, dtype=torch.float64)
        y = torch.randint(1, V, (1, max(U - 1, 1)), generator=g)
        want = oracle.rnnt_brute_force(z[0], y[0][: U - 1])
        got = oracle.rnnt_loss(z, y, [T], [U - 1], reduction="none")[0]
        assert abs(float(got) - float(want)) < 1e-12
    # padded batch entries do not leak into shorter utterances
    z = torch.randn(2, 5, 4, 6, generator=g, dtype=torch.float64)
    y = torch.randint(1, 6, (2, 3), generator=g)
    both = oracle.rnnt_loss(z, y, [5, 3], [3, 2], reduction="none")
    solo = oracle.rnnt_loss(z[1:, :3, :3], y[1:, :2], [3], [2], reduction="none")
    assert abs(float(both[1]) - float(solo[0])) < 1e-12


@pytest.mark.parametrize("name", ["transducer_rnn.npz", "transducer_gru.npz", "transducer_conformer.npz"])
def test_transducer(oracle, name):
    """encoder -> DecoderRNNT -> JointNetwork against the reference modules (e2e_asr_transducer.py:510-563);
    the loss value in the fixture comes from oracle.rnnt_loss itself (warprnnt_pytorch is absent)."""
    p, sd, grads = split_golden(load_golden(name))
    sd = req(sd)
    gru = "gru" in name
    if "rnn" in name:
        hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 1, [1, 1])
        dl = 2
    elif gru:
        hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 1, 1], vgg=False, gru=True,
                                       proj=False)
        dl = 2
    else:
        cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4)
        mask = oracle.non_pad_mask(p["ilens"]).unsqueeze(-2)
        hs, hmask = oracle.encoder(sd, "encoder.", p["xs"], mask, cfg, training=True)
        hlens = hmask.squeeze(1).sum(1).tolist()
        dl = 1
    close(hs, p["hs_pad"], rtol=1e-4, atol=1e-5)
    ys_in, target, ulens = oracle.rnnt_prepare(p["ys"])
    z = oracle.rnnt_decoder(sd, "dec.", hs, ys_in, dl, gru=gru)
    close(z, p["pred_pad"], rtol=1e-4, atol=1e-5)
    loss = oracle.rnnt_loss(z, target, hlens, ulens)
    assert abs(float(loss) - float(p["loss"])) < 1e-5 * abs(float(p["loss"]))
    loss.backward()
    _grad_check(sd, grads)


# ---- feature-side layers ---------------------------------------------------------
def _masked(feats, lens):
    return feats * (torch.arange(feats.shape[1]).view(1, -1, 1) < torch.as_tensor(lens).view(-1, 1, 1))


def test_feature_layers(oracle):
    """SpecAug (seeded CPU draws), GlobalMVN and UtteranceMVN against the espnet2 layers' own outputs"""
    p, _, _ = split_golden(load_golden("feature_layers.npz"))
    for tag in ("eq", "ragged"):
        lens = p["lens_%s" % tag].tolist()
        x = _masked(p["feats"], lens)
        torch.manual_seed(77)
        y = oracle.specaug(x.clone(), lens, 5, (0, 6), 2, (0, 20), 2)
        close(y, p["specaug_%s" % tag], rtol=1e-5, atol=1e-6)
        torch.manual_seed(78)
        y = oracle.specaug(x.clone(), lens, 5, (0, 6), 2, (0, 20), 2, apply=(False, True, True))
        assert torch.equal(y, p["specaug_nowarp_%s" % tag])           # masking alone is exact
        torch.manual_seed(79)
        y = oracle.specaug(x.clone(), lens, 7, apply=(True, False, False))
        close(y, p["specaug_warponly_%s" % tag], rtol=1e-5, atol=1e-6)
    lens = [120, 97, 64, 9]
    x = _masked(p["feats"], lens)
    cnt = float(p["stats_count"])
    mean = (p["stats_sum"].double() / cnt)
    std = torch.sqrt(torch.clamp(p["stats_sum_square"].double() / cnt - mean * mean, min=1e-20))
    for nm in (1, 0):
        for nv in (1, 0):
            close(oracle.global_mvn(x.clone(), lens, mean.float(), std.float(), bool(nm), bool(nv)), p["gmvn_%d%d" % (nm, nv)],
                  rtol=1e-5, atol=1e-6)
            close(oracle.utterance_mvn(x.clone(), lens, bool(nm), bool(nv)), p["umvn_%d%d" % (nm, nv)], rtol=1e-5, atol=1e-6)


def test_frontend(oracle):
    """8f rank 4: the numpy restatement of Stft -> power -> LogMel against the reference layers' recorded outputs
    (torch.stft), the mel matrix against the two values printed in librosa.filters.mel's docstring (the only
    known answers available without librosa), and the product's own mel matrix against the oracle's."""
    import numpy as np
    from espnet_amd.espnet2.frontend import mel_filterbank
    g = load_golden("frontend.npz")
    wav, wlens = np.asarray(g["wav"]), np.asarray(g["wlens"])
    for tag, kw in (("default", dict()), ("win400", dict(n_fft=512, win_length=400, hop=160, n_mels=40, htk=True)),
                    ("n256", dict(n_fft=256, hop=64, n_mels=23, fmin=80, fmax=7600))):
        feat, olens = oracle.logmel_frontend(wav, wlens, **kw)
        assert olens.tolist() == np.asarray(g[tag + "_flens"]).tolist()
        ref = np.asarray(g[tag + "_feats"])
        assert feat.shape == ref.shape
        err = np.abs(feat - ref).max()
        print(f"[oracle] frontend {tag}: max abs err {err:.2e} (range {ref.min():.1f}..{ref.max():.1f})")
        assert err < 2e-3
        s = oracle.stft(wav, kw.get("n_fft", 512), kw.get("hop", 128), kw.get("win_length"))
        sr = np.asarray(g[tag + "_stft"])
        T = s.shape[1]
        keep = (np.arange(T)[None, :] < olens[:, None])[..., None]
        assert np.abs(s.real * keep - sr[..., 0]).max() < 2e-4 * np.abs(sr).max()
        assert np.abs(s.imag * keep - sr[..., 1]).max() < 2e-4 * np.abs(sr).max()
    m = oracle.mel_filterbank(22050, 2048)
    assert m.shape == (128, 1025) and m[0, 0] == 0 and round(float(m[0, 1]), 3) == 0.016 and m[-1, -1] == 0
    assert round(float(oracle.mel_filterbank(22050, 2048, fmax=8000)[0, 1]), 2) == 0.02
    for kw in (dict(sr=16000, n_fft=512, n_mels=80), dict(sr=16000, n_fft=512, n_mels=40, htk=True),
               dict(sr=8000, n_fft=256, n_mels=23, fmin=80, fmax=3800)):
        a, b = oracle.mel_filterbank(**kw), mel_filterbank(**kw)
        assert np.abs(a - b).max() < 1e-7 * np.abs(a).max() + 1e-9


# ---- round 2: the width at which bf16 mode dispatches the fused attention kernels (d = 256, h = 4, d_k = 64) ----
def _seeded_sd(shapes, salt, oracle_dir=None):
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    return SW, {k: SW.seeded_value(k, shp, salt).requires_grad_(True) for k, shp in shapes.items()}


def _mha_shapes(d, rel):
    sh = {}
    for n in ("q", "k", "v", "out"):
        sh["linear_%s.weight" % n] = (d, d)
        sh["linear_%s.bias" % n] = (d,)
    if rel:
        sh["linear_pos.weight"] = (d, d)
        sh["pos_bias_u"] = (4, d // 4)
        sh["pos_bias_v"] = (4, d // 4)
    return sh


def _check_seeded_grads(SW, sd, fixture, tol, prefix=""):
    worst = 0.0
    for k, v in sd.items():
        if not (torch.is_tensor(v) and v.requires_grad):
            continue
        fx = {kk[len(prefix):] if prefix and kk.startswith(prefix) else kk: vv for kk, vv in fixture.items()} if prefix else fixture
        if not any(t + k in fx for t in ("grad/", "gprobe_r/")):
            continue
        assert v.grad is not None, k
        kind, e = SW.grad_check(k, v.grad, fx)
        worst = max(worst, e)
        assert e <= tol, (k, kind, e)
    return worst


def test_rel_mha_dk64(oracle):
    g = load_golden("rel_mha_dk64.npz")
    SW, sd = _seeded_sd(_mha_shapes(256, True), 71)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oracle.rel_mha(sd, "", x, torch.from_numpy(g["pos"]), torch.from_numpy(g["mask"]), 4)
    close(y, torch.from_numpy(g["y"]), rtol=1e-4, atol=2e-5)
    assert bool((y[2] == sd["linear_out.bias"]).all())        # the fully masked utterance: zeros through linear_out
    y.backward(torch.from_numpy(g["gy"]))
    close(x.grad, torch.from_numpy(g["gx"]), rtol=1e-4, atol=2e-5)
    assert _check_seeded_grads(SW, sd, g, 1e-4) < 1e-4


def test_mha_dk64(oracle):
    g = load_golden("mha_dk64.npz")
    SW, sd = _seeded_sd(_mha_shapes(256, False), 61)
    q = torch.from_numpy(g["q"]).requires_grad_(True)
    mem = torch.from_numpy(g["mem"]).requires_grad_(True)
    y = oracle.mha(sd, "", q, mem, mem, torch.from_numpy(g["mmask"]), 4)
    close(y, torch.from_numpy(g["y"]), rtol=1e-4, atol=2e-5)
    y.backward(torch.from_numpy(g["gy"]))
    close(q.grad, torch.from_numpy(g["gq"]), rtol=1e-4, atol=2e-5)
    close(mem.grad, torch.from_numpy(g["gmem"]), rtol=1e-4, atol=2e-5)
    assert _check_seeded_grads(SW, sd, g, 1e-4) < 1e-4
    q2 = torch.from_numpy(g["q"]).requires_grad_(True)
    y2 = oracle.mha(sd, "", q2, q2, q2, torch.from_numpy(g["cmask"]), 4)
    close(y2, torch.from_numpy(g["y_self"]), rtol=1e-4, atol=2e-5)


def test_e2e_conformer_dk64(oracle):
    """the oracle end to end at adim 256 / aheads 4 against the reference's loss, encoder output, gradients and
    greedy ids (weights from oracle/seeded_weights.py on both sides)"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    from conftest import e2e_dk64_model
    g = load_golden("e2e_conformer_dk64.npz")
    model, cfg = e2e_dk64_model()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in model.state_dict().items()}
    xs, ilens, ys = torch.from_numpy(g["xs"]), g["ilens"].tolist(), torch.from_numpy(g["ys"])
    out = oracle.e2e_forward(sd, xs, ilens, ys, cfg, training=True)
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    assert abs(float(out["loss_ctc"]) - float(g["loss_ctc"])) <= 2e-5 * abs(float(g["loss_ctc"]))
    assert abs(float(out["acc"]) - float(g["acc"])) < 1e-6
    close(out["hs_pad"], torch.from_numpy(g["hs_pad"]), rtol=2e-4, atol=2e-5)
    out["loss"].backward()
    assert _check_seeded_grads(SW, sd, g, 2e-4) < 2e-4


def test_postnorm_layers(oracle):
    """normalize_before=False and / or concat_after=True (conformer/encoder_layer.py:99-157, transformer/encoder_layer.py:53-101,
    decoder_layer.py:60-134): the oracle's layers against the reference's outputs and gradients (postnorm_layers.npz)"""
    from conftest import POSTNORM_VARIANTS, postnorm_layer
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in load_golden("postnorm_layers.npz").items()}
    for tag, nb, cat in POSTNORM_VARIANTS:
        cfg = dict(aheads=4, activation="swish", normalize_before=nb, concat_after=cat)
        for kind in ("conf", "trf", "dec"):
            sd = req({k: v.detach() for k, v in postnorm_layer(kind, nb, cat).state_dict().items()})
            pre = "%s_%s/" % (kind, tag)
            if kind == "dec":
                tgt, mem = g["tgt"].clone().requires_grad_(True), g["x"].clone().requires_grad_(True)
                y = oracle.decoder_layer(sd, "", tgt, g["tmask"], mem, g["mask"], 4, nb, cat)
                close(y, g[pre + "y"])
                y.backward(g["gyt"])
                close(tgt.grad, g[pre + "gtgt"])
                close(mem.grad, g[pre + "gmem"])
            else:
                x = g["x"].clone().requires_grad_(True)
                y = (oracle.conformer_layer(sd, "", x, g["pos"], g["mask"], cfg, True) if kind == "conf"
                     else oracle.transformer_enc_layer(sd, "", x, g["mask"], cfg))
                close(y, g[pre + "y"])
                y.backward(g["gy"])
                close(x.grad, g[pre + "gx"])
            check_param_grads(sd, {k[len(pre) + 5:]: v for k, v in g.items() if k.startswith(pre + "grad/")})


def test_e2e_conformer_d512(oracle):
    """the oracle end to end at the width of the reference's large recipes (adim 512, aheads 8, eunits = dunits = 2048;
    oracle/gen_golden_r4b.py) against the reference's loss, encoder output and gradients"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    from conftest import e2e_d512_model
    g = load_golden("e2e_conformer_d512.npz")
    model, cfg = e2e_d512_model()
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
          for k, v in model.state_dict().items()}
    xs, ilens, ys = torch.from_numpy(g["xs"]), g["ilens"].tolist(), torch.from_numpy(g["ys"])
    out = oracle.e2e_forward(sd, xs, ilens, ys, cfg, training=True)
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    assert abs(float(out["loss_ctc"]) - float(g["loss_ctc"])) <= 2e-5 * abs(float(g["loss_ctc"]))
    assert abs(float(out["acc"]) - float(g["acc"])) < 1e-6
    close(out["hs_pad"], torch.from_numpy(g["hs_pad"]), rtol=2e-4, atol=2e-5)
    out["loss"].backward()
    assert _check_seeded_grads(SW, sd, g, 2e-4) < 2e-4


# ---- round-3 fixtures (oracle/gen_golden_r3.py): the remaining get_activation entries, any-width subsampling, scheduled
# sampling, the WarmupLR / Adadelta traces ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["hardtanh", "tanh", "selu"])
def test_ffn_other_activations(oracle, name):
    p, sd, grads = split_golden(load_golden("ffn_%s.npz" % name))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.ffn(sd, "", x, oracle.activation(name))
    close(y, p["y"])
    y.backward(p["gy"])
    close(x.grad, p["gx"])
    check_param_grads(sd, grads)


def test_conv_module_selu(oracle):
    p, sd, grads = split_golden(load_golden("conv_module_selu.npz"))
    sd = req(sd)
    x = p["x"].clone().requires_grad_(True)
    y = oracle.conv_module(sd, "", x, oracle.activation("selu"), True)
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    y.backward(p["gy"])
    close(x.grad, p["gx"], rtol=1e-4, atol=1e-5)
    check_param_grads(sd, grads, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["subsampling_odim40.npz", "subsampling6_odim48.npz"])
def test_subsampling_any_width(oracle, name):
    p, sd, grads = split_golden(load_golden(name))
    sd = req(sd)
    y, _, m = oracle.conv2d_subsampling(sd, "", p["x"], p["mask"], False)
    close(y, p["y"], rtol=1e-4, atol=1e-5)
    assert torch.equal(m, p["ymask"])
    y.backward(p["gy"])
    check_param_grads(sd, grads, rtol=1e-4, atol=1e-4)


def test_e2e_rnn_scheduled_sampling(oracle):
    """rnn/decoders.py:249-254: the decoder feeds back its own argmax token when random.random() < sampling_probability"""
    import random
    p, sd, grads = split_golden(load_golden("e2e_rnn_ss.npz"))
    sd = req(sd)
    hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 1, 1])
    loss_ctc = oracle.ctc_loss(oracle.linear(sd, "ctc.ctc_lo.", hs), torch.tensor(hlens), p["ys"])
    random.seed(7)
    loss_att, acc, _ = oracle.rnn_att_decoder(sd, "dec.", hs, hlens, p["ys"], 6, 6, 2, "att.0.", sampling_probability=0.5)
    close(loss_att.detach(), p["loss_att"], rtol=1e-4, atol=1e-5)
    assert abs(acc - float(p["acc"])) < 1e-6
    loss = 0.5 * loss_ctc + 0.5 * loss_att
    close(loss.detach(), p["loss"], rtol=1e-4, atol=1e-5)
    loss.backward()
    _grad_check(sd, grads)
    assert sum(c < 0.5 for c in p["coins"].tolist()[:6]) >= 1       # the fixture does exercise the sampling branch


def test_e2e_rnn_vggblstm(oracle):
    """BASELINE config 4's encoder type: VGG + stacked non-projected BLSTM + tanh(l_last) (rnn/encoders.py:103-162)"""
    p, sd, grads = split_golden(load_golden("e2e_rnn_vggblstm.npz"))
    sd = req(sd)
    hs, hlens = oracle.rnn_encoder(sd, "enc.", p["xs"], p["ilens"].tolist(), 2, [1, 1, 1], proj=False)
    assert hlens == p["hlens"].tolist()
    close(hs, p["hs_pad"], rtol=1e-4, atol=1e-5)
    loss_ctc = oracle.ctc_loss(oracle.linear(sd, "ctc.ctc_lo.", hs), torch.tensor(hlens), p["ys"])
    loss_att, acc, _ = oracle.rnn_att_decoder(sd, "dec.", hs, hlens, p["ys"], 6, 6, 2, "att.0.")
    loss = 0.5 * loss_ctc + 0.5 * loss_att
    close(loss.detach(), p["loss"], rtol=1e-4, atol=1e-5)
    assert abs(acc - float(p["acc"])) < 1e-6
    loss.backward()
    _grad_check(sd, grads)


# ---- round 4: the oracle's decoding functions (cached decoder step, CTC prefix scores, beam search) ------------------------------
def _nbest(g, tag):
    lens, flat = np.asarray(g[tag + "_lens"]).tolist(), np.asarray(g[tag + "_yseq"]).tolist()
    seqs, o = [], 0
    for n in lens:
        seqs.append(flat[o:o + n])
        o += n
    return seqs, np.asarray(g[tag + "_scores"]).tolist()


@pytest.mark.parametrize("name,cfg", [
    ("e2e_conformer.npz", dict(conformer=True, rel_pos=True, activation="swish")),
    ("e2e_transformer.npz", dict(conformer=False, rel_pos=False)),
])
def test_oracle_beam_search_small(oracle, name, cfg):
    """oracle.beam_search (BeamSearch semantics) against the reference's recorded 3-best: ctc_weight 0 / 0.3 / 1, beam 4, |V| = 50"""
    p, sd, _ = split_golden(load_golden(name))
    cfg = dict(cfg, aheads=4, mtlalpha=0.3, lsm_weight=0.1, odim=50)
    with torch.no_grad():
        if cfg["conformer"]:     # the fixture decoded after one training forward: BatchNorm running statistics moved
            bn = {}
            oracle.e2e_forward(sd, p["xs"], p["ilens"].tolist(), p["ys"], cfg, training=True, bn_state=bn)
            sd = dict(sd, **bn)
        hs, _ = oracle.encoder(sd, "encoder.", p["xs"][1:2, :77], None, cfg, training=False)
        for cw in (0.0, 0.3, 1.0):
            tag = "beam_w%02d" % int(cw * 10)
            nb = oracle.beam_search(sd, hs[0], cfg, dict(decoder=1 - cw, ctc=cw, length_bonus=0.2), 4, 0.0)[:3]
            seqs, scores = _nbest(p, tag)
            assert [h["yseq"] for h in nb] == seqs, tag
            for h, s in zip(nb, scores):
                assert abs(h["score"] - s) <= 1e-4 * max(1.0, abs(s)), (tag, h["score"], s)


def test_oracle_decode_c2width(oracle):
    """the decoding oracle at BASELINE config 2's width (decode_c2width.npz: d 256, |V| 5000, beam 10): encoder output and CTC
    posteriors of the 300-frame utterance, then hybrid searches in both semantics - "ids" = the reference's BeamSearch, "full" =
    its BatchBeamSearch (they differ on the 1000-frame utterance, where <eos> wins from outside the pre-beam) - ids exact,
    total and per-scorer scores 1e-4."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    from espnet_amd.nets.e2e_asr_conformer import E2E
    g = load_golden("decode_c2width.npz")
    model = SW.decode_r4_model(E2E)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4, odim=SW.DECODE_R4["odim"])
    xs = SW.decode_r4_inputs()
    torch.set_num_threads(8)
    with torch.no_grad():
        for u, cases in ((2, [(0.3, 0.2, 0.1), (0.3, 0.0, 0.0), (0.0, 0.0, 0.0)]), (0, [(0.3, 0.0, 0.0)])):
            hs, _ = oracle.encoder(sd, "encoder.", xs[u].unsqueeze(0), None, cfg, training=False)
            close(hs[0, ::8], torch.from_numpy(g["u%d_enc" % u]), rtol=2e-4, atol=2e-5)
            logp = torch.log_softmax(oracle.linear(sd, "ctc.ctc_lo.", hs[0]), -1)
            close(logp[::8, ::50], torch.from_numpy(g["u%d_logp" % u]), rtol=2e-4, atol=2e-4)
            assert logp.argmax(-1).tolist() == g["u%d_ctc_argmax" % u].tolist()
            for cw, ratio, pen in cases:
                for mode, nm in (("ids", "beam"), ("full", "bbeam")):
                    tag = "u%d_%s_w%02d_r%02d" % (u, nm, int(cw * 10), int(ratio * 10))
                    nb = oracle.beam_search(sd, hs[0], cfg, dict(decoder=1 - cw, ctc=cw, length_bonus=pen), SW.DECODE_R4["beam"],
                                            ratio, mode=mode)
                    seqs, scores = _nbest(g, tag)
                    assert len(nb) == int(g[tag + "_nended"]), (tag, len(nb))
                    assert [h["yseq"] for h in nb[:len(seqs)]] == seqs, tag
                    for k, (h, s) in enumerate(zip(nb, scores)):
                        assert abs(h["score"] - s) <= 1e-4 * max(1.0, abs(s)), (tag, h["score"], s)
                        for kk in h["scores"]:
                            ref = float(g[tag + "_sc_" + kk][k])
                            assert abs(h["scores"][kk] - ref) <= 1e-4 * max(1.0, abs(ref)), (tag, kk)
    seq_a, _ = _nbest(g, "u0_beam_w03_r00")
    seq_b, _ = _nbest(g, "u0_bbeam_w03_r00")
    assert seq_a != seq_b            # the fixture does separate the two semantics


def test_oracle_decode_c2width_long(oracle):
    """... on a 657-frame memory (decode_c2width_long.npz, oracle/gen_golden_r4c.py: encoder outputs of utterances 0, 1, 0 back to
    back): oracle.beam_search in "ids" mode against the reference's BeamSearch - ids exact, scores 1e-4"""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    from espnet_amd.nets.e2e_asr_conformer import E2E
    g = load_golden("decode_c2width_long.npz")
    model = SW.decode_r4_model(E2E)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4, odim=SW.DECODE_R4["odim"])
    xs = SW.decode_r4_inputs()
    torch.set_num_threads(8)
    with torch.no_grad():
        hs = [oracle.encoder(sd, "encoder.", xs[u].unsqueeze(0), None, cfg, training=False)[0][0] for u in (0, 1)]
        enc = torch.cat([hs[0], hs[1], hs[0]], 0)
        close(enc[::16], torch.from_numpy(g["enc_sample"]), rtol=2e-4, atol=2e-5)
        nb = oracle.beam_search(sd, enc, cfg, dict(decoder=0.7, ctc=0.3, length_bonus=0.1), SW.DECODE_R4["beam"], 0.2, mode="ids")
    seqs, scores = _nbest(g, "long_beam")
    assert [h["yseq"] for h in nb[:len(seqs)]] == seqs
    for h, s_ in zip(nb, scores):
        assert abs(h["score"] - s_) <= 1e-4 * max(1.0, abs(s_)), (h["score"], s_)
