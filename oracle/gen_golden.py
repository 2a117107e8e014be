#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU).

Build-container only: needs /root/reference on disk; nothing here travels to the GPU box except the
.npz data files it writes (inputs, seeded weights, reference outputs and gradients).  Missing
third-party packages that the hot path never computes with are replaced by inert in-process stubs
(SURVEY.md §8c).  Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch


def install_stubs():
    class _Permissive(types.ModuleType):
        """inert stand-in: any attribute is a do-nothing callable / base class"""

        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return type(item, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})

    def mod(name, **attrs):
        m = _Permissive(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "chainer" not in sys.modules:
        class _Rep:
            def report(self, *a, **k):
                pass

        class _Chain:
            pass

        ch = mod("chainer", Chain=_Chain)
        ch.reporter = mod("chainer.reporter", report=lambda *a, **k: None, Reporter=_Rep)
        ch.training = mod("chainer.training")
        ch.training.extension = mod("chainer.training.extension", Extension=object)
        ch.datasets = mod("chainer.datasets")
    if "editdistance" not in sys.modules:
        def _eval(a, b):
            a, b = list(a), list(b)
            d = list(range(len(b) + 1))
            for i in range(1, len(a) + 1):
                prev, d[0] = d[0], i
                for j in range(1, len(b) + 1):
                    cur = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
                    prev, d[j] = d[j], cur
            return d[-1]
        mod("editdistance", eval=_eval)
    for name in ("configargparse", "humanfriendly", "librosa", "torch_complex", "pytorch_wpe"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:  # noqa: BLE001
                m = mod(name)
                if name == "configargparse":
                    m.ArgumentParser = argparse.ArgumentParser
                    m.YAMLConfigFileParser = object
                    m.ArgumentDefaultsHelpFormatter = argparse.ArgumentDefaultsHelpFormatter
                if name == "torch_complex":
                    m.tensor = mod("torch_complex.tensor", ComplexTensor=object)
                    m.functional = mod("torch_complex.functional")
    if "typeguard" not in sys.modules:
        try:
            import typeguard  # noqa: F401
        except Exception:  # noqa: BLE001
            mod("typeguard", check_argument_types=lambda *a, **k: True, check_return_type=lambda *a, **k: True,
                typechecked=lambda f=None, **k: f if f else (lambda g: g))
    if not hasattr(np, "int"):
        np.int = int  # nets_utils.py:406 uses the alias removed in numpy 1.24
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "bool"):
        np.bool = bool


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad/"):
    return {prefix + k: p.grad.detach().cpu().numpy() for k, p in module.named_parameters() if p.grad is not None}


def save(path, **arrs):
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    a = ap.parse_args()
    install_stubs()
    sys.path.insert(0, a.ref)
    os.makedirs(a.out, exist_ok=True)
    out = lambda n: os.path.join(a.out, n)  # noqa: E731
    torch.set_num_threads(4)

    from espnet.nets.pytorch_backend.conformer.convolution import ConvolutionModule
    from espnet.nets.pytorch_backend.conformer.encoder_layer import EncoderLayer as ConfLayer
    from espnet.nets.pytorch_backend.conformer.swish import Swish
    from espnet.nets.pytorch_backend.ctc import CTC
    from espnet.nets.pytorch_backend.transformer.attention import (MultiHeadedAttention,
                                                                    RelPositionMultiHeadedAttention)
    from espnet.nets.pytorch_backend.transformer.decoder import Decoder
    from espnet.nets.pytorch_backend.transformer.embedding import (PositionalEncoding, RelPositionalEncoding)
    from espnet.nets.pytorch_backend.transformer.label_smoothing_loss import LabelSmoothingLoss
    from espnet.nets.pytorch_backend.transformer.layer_norm import LayerNorm
    from espnet.nets.pytorch_backend.transformer.mask import subsequent_mask
    from espnet.nets.pytorch_backend.transformer.positionwise_feed_forward import PositionwiseFeedForward
    from espnet.nets.pytorch_backend.transformer.subsampling import Conv2dSubsampling
    from espnet.nets.pytorch_backend.nets_utils import th_accuracy

    # ---- a10 LayerNorm ---------------------------------------------------------------------
    torch.manual_seed(10)
    ln = LayerNorm(64)
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.uniform_(-0.5, 0.5)
    x = torch.randn(3, 7, 64, requires_grad=True)
    y = ln(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    save(out("layernorm.npz"), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, **sd_np(ln, "sd/"), **grads_np(ln))

    # ---- a7 rel_shift on T1 != T2, and RelPositionMultiHeadedAttention --------------------
    torch.manual_seed(7)
    att = RelPositionMultiHeadedAttention(4, 64, 0.0)
    xs = torch.randn(2, 3, 5, 9)
    xs2 = torch.randn(1, 2, 6, 6)
    save(out("rel_shift.npz"), x=xs, y=att.rel_shift(xs), x2=xs2, y2=att.rel_shift(xs2))
    pe = RelPositionalEncoding(64, 0.0)
    x = torch.randn(2, 11, 64, requires_grad=True)
    _, pos = pe(x)
    mask = torch.ones(2, 1, 11, dtype=torch.bool)
    mask[1, 0, 8:] = False
    y = att(x, x, x, pos, mask)
    gy = torch.randn_like(y)
    y.backward(gy)
    save(out("rel_mha.npz"), x=x.detach(), pos=pos.detach(), mask=mask, y=y.detach(), gy=gy, gx=x.grad,
         attn=att.attn.detach(), **sd_np(att, "sd/"), **grads_np(att))

    # ---- a6 MultiHeadedAttention: self (causal mask) and source attention, fully masked row --
    torch.manual_seed(6)
    att = MultiHeadedAttention(4, 64, 0.0)
    q = torch.randn(2, 5, 64, requires_grad=True)
    mem = torch.randn(2, 9, 64, requires_grad=True)
    mmask = torch.ones(2, 1, 9, dtype=torch.bool)
    mmask[1, 0, 6:] = False
    y = att(q, mem, mem, mmask)
    gy = torch.randn_like(y)
    y.backward(gy)
    cm = subsequent_mask(5).unsqueeze(0).expand(2, 5, 5).clone()
    cm[1, 2, :] = False   # a fully masked query row -> zeros after masked_fill(0)
    y2 = att(q.detach(), q.detach(), q.detach(), cm)
    save(out("mha.npz"), q=q.detach(), mem=mem.detach(), mmask=mmask, y=y.detach(), gy=gy, gq=q.grad, gmem=mem.grad,
         cmask=cm, y_self=y2.detach(), **sd_np(att, "sd/"), **grads_np(att))

    # ---- a8 FFN (swish and relu) -----------------------------------------------------------
    torch.manual_seed(8)
    for name, actm in (("swish", Swish()), ("relu", torch.nn.ReLU())):
        ff = PositionwiseFeedForward(64, 96, 0.0, actm)
        x = torch.randn(2, 6, 64, requires_grad=True)
        y = ff(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        save(out("ffn_%s.npz" % name), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, **sd_np(ff, "sd/"),
             **grads_np(ff))

    # ---- a9 ConvolutionModule train + eval ---------------------------------------------------
    torch.manual_seed(9)
    cm_ = ConvolutionModule(64, 7, Swish())
    cm_.norm.weight.data.uniform_(0.5, 1.5)
    cm_.norm.bias.data.uniform_(-0.3, 0.3)
    sd0 = sd_np(cm_, "sd/")
    x = torch.randn(3, 13, 64, requires_grad=True)
    cm_.train()
    y = cm_(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    sd1 = sd_np(cm_, "sd_after/")
    cm_.eval()
    ye = cm_(x.detach())
    save(out("conv_module.npz"), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, y_eval=ye.detach(), **sd0, **sd1,
         **grads_np(cm_))

    # ---- a3/a4 Conv2dSubsampling with abs and rel positional encoding -------------------------
    torch.manual_seed(3)
    for name, pcls in (("abs", PositionalEncoding), ("rel", RelPositionalEncoding)):
        sub = Conv2dSubsampling(20, 64, 0.0, pcls(64, 0.0))
        x = torch.randn(2, 37, 20)
        m = torch.ones(2, 1, 37, dtype=torch.bool)
        m[1, 0, 29:] = False
        y, ym = sub(x, m)
        pos = None
        if isinstance(y, tuple):
            y, pos = y
        gy = torch.randn_like(y)
        y.backward(gy)
        extra = dict(pos=pos.detach()) if pos is not None else {}
        save(out("subsampling_%s.npz" % name), x=x, mask=m, y=y.detach(), ymask=ym, gy=gy, **extra, **sd_np(sub, "sd/"),
             **grads_np(sub))

    # ---- a5 conformer encoder layer, the four variants of test_e2e_asr_conformer.py:69-88 ----
    torch.manual_seed(5)
    for macaron in (False, True):
        for cnn in (False, True):
            lay = ConfLayer(64, RelPositionMultiHeadedAttention(4, 64, 0.0),
                            PositionwiseFeedForward(64, 96, 0.0, Swish()),
                            PositionwiseFeedForward(64, 96, 0.0, Swish()) if macaron else None,
                            ConvolutionModule(64, 7, Swish()) if cnn else None, 0.0, True, False)
            sd0 = sd_np(lay, "sd/")
            x = torch.randn(2, 12, 64, requires_grad=True)
            _, pos = RelPositionalEncoding(64, 0.0)(x)
            m = torch.ones(2, 1, 12, dtype=torch.bool)
            m[0, 0, 9:] = False
            (y, _), _ = lay((x, pos), m)
            gy = torch.randn_like(y)
            y.backward(gy)
            save(out("conformer_layer_m%d_c%d.npz" % (macaron, cnn)), x=x.detach(), pos=pos.detach(), mask=m,
                 y=y.detach(), gy=gy, gx=x.grad, **sd0, **grads_np(lay))

    # ---- a14 LabelSmoothingLoss both normalisations + th_accuracy ----------------------------
    torch.manual_seed(14)
    x = torch.randn(3, 5, 17, requires_grad=True)
    t = torch.randint(0, 17, (3, 5))
    t[0, 3:] = -1
    t[2, 4:] = -1
    res = {}
    for norm in (False, True):
        x.grad = None
        loss = LabelSmoothingLoss(17, -1, 0.1, norm)(x, t)
        loss.backward()
        res["loss_n%d" % norm] = loss.detach()
        res["gx_n%d" % norm] = x.grad.clone()
    x.grad = None
    l0 = LabelSmoothingLoss(17, -1, 0.0, False)(x, t)
    save(out("lsm_loss.npz"), x=x.detach(), t=t, acc=th_accuracy(x.detach().view(-1, 17), t, -1), loss_s0=l0.detach(),
         **res)

    # ---- a15 CTC: lengths of test/test_loss.py:18-20, repeated labels, infeasible ------------
    torch.manual_seed(15)
    ctc = CTC(6, 8, 0.0, ctc_type="builtin", reduce=True)
    hs = torch.randn(3, 5, 8, requires_grad=True)
    hlens = torch.tensor([5, 4, 5])
    ys = torch.tensor([[1, 2, -1], [3, 3, 4], [2, -1, -1]])     # utt1: repeated label, T=4 >= 2+1+... feasible
    loss = ctc(hs, hlens, ys)
    loss.backward()
    logits = ctc.ctc_lo(hs).detach()
    # gradient wrt the raw activations (what warp-ctc returns) for direct kernel checks
    lg = logits.clone().requires_grad_(True)
    l2 = torch.nn.functional.ctc_loss(lg.transpose(0, 1).log_softmax(2), torch.tensor([1, 2, 3, 3, 4, 2], dtype=torch.int32),
                                      torch.tensor([5, 4, 5], dtype=torch.int32), torch.tensor([2, 3, 1], dtype=torch.int32),
                                      reduction="sum") / 3
    l2.backward()
    # infeasible: 3 labels with a repeat need T >= 4, give T = 3
    linf = torch.nn.functional.ctc_loss(logits[:1, :3].transpose(0, 1).log_softmax(2), torch.tensor([3, 3, 4], dtype=torch.int32),
                                        torch.tensor([3], dtype=torch.int32), torch.tensor([3], dtype=torch.int32),
                                        reduction="sum")
    # per-utterance nll
    nll = torch.nn.functional.ctc_loss(logits.transpose(0, 1).log_softmax(2), torch.tensor([1, 2, 3, 3, 4, 2], dtype=torch.int32),
                                       torch.tensor([5, 4, 5], dtype=torch.int32), torch.tensor([2, 3, 1], dtype=torch.int32),
                                       reduction="none")
    save(out("ctc.npz"), hs=hs.detach(), hlens=hlens, ys=ys, loss=loss.detach(), ghs=hs.grad, logits=logits,
         glogits=lg.grad, loss_direct=l2.detach(), loss_infeasible=linf.detach(), nll=nll.detach(),
         argmax=ctc.argmax(hs).detach(), log_softmax=ctc.log_softmax(hs).detach(), **sd_np(ctc, "sd/"), **grads_np(ctc))

    # ---- a12 Decoder: full forward/backward + cached one-step --------------------------------
    torch.manual_seed(12)
    dec = Decoder(odim=23, attention_dim=64, attention_heads=4, linear_units=96, num_blocks=2, dropout_rate=0.0,
                  positional_dropout_rate=0.0, self_attention_dropout_rate=0.0, src_attention_dropout_rate=0.0)
    ys_in = torch.randint(1, 22, (2, 6))
    mem = torch.randn(2, 9, 64, requires_grad=True)
    mmask = torch.ones(2, 1, 9, dtype=torch.bool)
    mmask[1, 0, 7:] = False
    tmask = subsequent_mask(6).unsqueeze(0).expand(2, 6, 6)
    y, _ = dec(ys_in, tmask, mem, mmask)
    gy = torch.randn_like(y)
    y.backward(gy)
    dec.eval()
    cache = None
    steps = []
    for i in range(1, 5):
        lp, cache = dec.forward_one_step(ys_in[:1, :i], subsequent_mask(i).unsqueeze(0), mem[:1].detach(), cache=cache)
        steps.append(lp.detach())
    save(out("decoder.npz"), ys_in=ys_in, mem=mem.detach(), mmask=mmask, y=y.detach(), gy=gy, gmem=mem.grad,
         step_logp=torch.stack(steps), **sd_np(dec, "sd/"), **grads_np(dec))

    # ---- a1 end-to-end: small Conformer and small Transformer (BASELINE config 1) -------------
    from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E as ConfE2E
    from espnet.nets.pytorch_backend.e2e_asr_transformer import E2E as TrfE2E

    def e2e_case(name, cls, extra, seed):
        torch.manual_seed(seed)
        kw = dict(adim=64, aheads=4, elayers=2, eunits=128, dlayers=1, dunits=128, mtlalpha=0.3,
                  lsm_weight=0.1, dropout_rate=0.0, transformer_attn_dropout_rate=0.0,
                  transformer_length_normalized_loss=False, transformer_init="pytorch",
                  transformer_input_layer="conv2d", ctc_type="builtin", report_cer=False,
                  report_wer=False, char_list=None, sym_space="<space>", sym_blank="<blank>")
        kw.update(extra)
        ns = argparse.Namespace(**kw)
        model = cls(20, 50, ns)
        model.train()
        sd0 = sd_np(model, "sd/")
        g = torch.Generator().manual_seed(1)
        xs = torch.randn(2, 100, 20, generator=g)
        ilens = torch.tensor([100, 77])
        ys = torch.randint(1, 49, (2, 9), generator=g)
        ys[1, 6:] = -1
        loss = model(xs, ilens, ys)
        loss.backward()
        rep = dict(loss=float(loss), acc=float(model.acc), hs=model.hs_pad.detach().clone(),
                   pred=model.pred_pad.detach().clone(), loss_ctc=float(model.ctc.loss))
        model.eval()
        with torch.no_grad():
            lz = model.ctc.argmax(model.encoder(xs[:1], None)[0])
            from itertools import groupby
            greedy = [v for v in (k[0] for k in groupby(lz[0].tolist())) if v != 0]
            model(xs, ilens, ys)   # eval-mode loss (BatchNorm running stats)
            eval_loss = float(model.loss)
            # a19: joint CTC/attention beam search n-best (reference BeamSearch, per-hypothesis scoring)
            from espnet.nets.beam_search import BeamSearch
            from espnet.nets.scorers.length_bonus import LengthBonus
            enc = model.encode(xs[1, :77].numpy())
            beam = {}
            for cw in (0.0, 0.3, 1.0):
                scorers = model.scorers()
                scorers["length_bonus"] = LengthBonus(50)
                bs = BeamSearch(beam_size=4, vocab_size=50, weights=dict(decoder=1.0 - cw, ctc=cw, length_bonus=0.2),
                                scorers=scorers, sos=model.sos, eos=model.eos, token_list=None,
                                pre_beam_score_key=None if cw == 1.0 else "full")
                nb = bs(x=enc, maxlenratio=0.0, minlenratio=0.0)[:3]
                tag = "beam_w%02d" % int(cw * 10)
                beam[tag + "_scores"] = np.asarray([float(h.score) for h in nb], dtype=np.float64)
                beam[tag + "_lens"] = np.asarray([len(h.yseq) for h in nb], dtype=np.int64)
                beam[tag + "_yseq"] = np.asarray(sum([[int(t) for t in h.yseq] for h in nb], []), dtype=np.int64)
        save(out(name), xs=xs, ilens=ilens, ys=ys, hs_pad=rep["hs"], pred_pad=rep["pred"],
             loss=rep["loss"], loss_ctc=rep["loss_ctc"], acc=rep["acc"], eval_loss=eval_loss, **beam, greedy=np.asarray(greedy, dtype=np.int64),
             **sd0, **grads_np(model))
        return model

    # ---- a2 espnet2 surface: ESPnetASRModel(ConformerEncoder, TransformerDecoder, CTC) ------------
    from espnet2.asr.ctc import CTC as CTC2
    from espnet2.asr.decoder.transformer_decoder import TransformerDecoder
    from espnet2.asr.encoder.conformer_encoder import ConformerEncoder
    from espnet2.asr.espnet_model import ESPnetASRModel
    torch.manual_seed(23)
    enc = ConformerEncoder(20, output_size=64, attention_heads=4, linear_units=96, num_blocks=2, dropout_rate=0.0,
                           positional_dropout_rate=0.0, attention_dropout_rate=0.0, macaron_style=True,
                           cnn_module_kernel=7)
    dec = TransformerDecoder(30, 64, attention_heads=4, linear_units=96, num_blocks=1, dropout_rate=0.0,
                             positional_dropout_rate=0.0, self_attention_dropout_rate=0.0,
                             src_attention_dropout_rate=0.0)
    m2 = ESPnetASRModel(vocab_size=30, token_list=["<blank>"] + [str(i) for i in range(1, 28)] + ["<space>", "<sos/eos>"], frontend=None, specaug=None,
                        normalize=None, encoder=enc, decoder=dec, ctc=CTC2(30, 64, ctc_type="builtin"),
                        rnnt_decoder=None, ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False)
    m2.train()
    sd0 = sd_np(m2, "sd/")
    g = torch.Generator().manual_seed(2)
    speech = torch.randn(3, 61, 20, generator=g)
    slen = torch.tensor([61, 50, 33])
    text = torch.randint(1, 29, (3, 7), generator=g)
    tlen = torch.tensor([7, 5, 3])
    for i, n in enumerate(tlen.tolist()):
        text[i, n:] = -1
    loss, stats, weight = m2(speech, slen, text, tlen)
    loss.backward()
    save(out("espnet2_model.npz"), speech=speech, speech_lengths=slen, text=text, text_lengths=tlen,
         loss=loss.detach(), loss_att=stats["loss_att"], loss_ctc=stats["loss_ctc"], acc=float(stats["acc"]),
         weight=weight, **sd0, **grads_np(m2))

    # ---- 8f rank 4: conv1d positionwise layers (MultiLayeredConv1d / Conv1dLinear) inside the espnet2 encoders ----
    from espnet2.asr.encoder.transformer_encoder import TransformerEncoder as TrfEncoder2
    for tag, cls, kw in (("conf_conv1d", ConformerEncoder, dict(positionwise_layer_type="conv1d", macaron_style=True,
                                                               cnn_module_kernel=7)),
                         ("conf_conv1dlin", ConformerEncoder, dict(positionwise_layer_type="conv1d-linear",
                                                                  positionwise_conv_kernel_size=5, use_cnn_module=False)),
                         ("trf_conv1d", TrfEncoder2, dict(positionwise_layer_type="conv1d",
                                                          positionwise_conv_kernel_size=3)),
                         ("conf_conv2d8", ConformerEncoder, dict(input_layer="conv2d8", use_cnn_module=False)),
                         ("trf_conv2d8", TrfEncoder2, dict(input_layer="conv2d8")),
                         ("conf_conv2d6", ConformerEncoder, dict(input_layer="conv2d6", use_cnn_module=False)),
                         ("trf_conv2d6", TrfEncoder2, dict(input_layer="conv2d6"))):
        torch.manual_seed(31)
        enc_pw = cls(20, output_size=64, attention_heads=4, linear_units=96, num_blocks=2, dropout_rate=0.0,
                     positional_dropout_rate=0.0, attention_dropout_rate=0.0, **kw)
        enc_pw.train()
        sd_pw = sd_np(enc_pw, "sd/")
        gp = torch.Generator().manual_seed(5)
        xs_pw = torch.randn(2, 61, 20, generator=gp)
        il_pw = torch.tensor([61, 44])
        y_pw, ol_pw, _ = enc_pw(xs_pw, il_pw)
        gy_pw = torch.randn(y_pw.shape, generator=gp)
        (y_pw * gy_pw).sum().backward()
        save(out("pw_%s.npz" % tag), xs=xs_pw, ilens=il_pw, y=y_pw.detach(), olens=ol_pw, gy=gy_pw, **sd_pw,
             **grads_np(enc_pw))

    # ---- espnet2 RNN encoders (RNNEncoder, VGGRNNEncoder) ----------------------------------------------------------------
    from espnet2.asr.encoder.rnn_encoder import RNNEncoder
    from espnet2.asr.encoder.vgg_rnn_encoder import VGGRNNEncoder
    for tag, cls, kw in (("rnnp", RNNEncoder, dict(num_layers=3, hidden_size=12, output_size=10, subsample=(2, 1))),
                         ("gru", RNNEncoder, dict(rnn_type="gru", bidirectional=False, use_projection=False, num_layers=2,
                                                  hidden_size=12, output_size=10, subsample=None)),
                         ("vgg", VGGRNNEncoder, dict(num_layers=1, hidden_size=12, output_size=10))):
        torch.manual_seed(37)
        enc_r = cls(20, **kw)
        enc_r.train()
        sd_r = sd_np(enc_r, "sd/")
        gp = torch.Generator().manual_seed(6)
        xs_r = torch.randn(3, 41, 20, generator=gp)
        il_r = torch.tensor([41, 30, 17])
        xs_r = xs_r * (torch.arange(41).view(1, -1, 1) < il_r.view(-1, 1, 1))
        y_r, ol_r, _ = enc_r(xs_r, il_r)
        gy_r = torch.randn(y_r.shape, generator=gp)
        (y_r * gy_r).sum().backward()
        save(out("enc2_%s.npz" % tag), xs=xs_r, ilens=il_r, y=y_r.detach(), olens=torch.as_tensor(ol_r), gy=gy_r, **sd_r,
             **grads_np(enc_r))

    # ---- espnet2 RNN model: RNNEncoder + attention RNNDecoder + CTC inside ESPnetASRModel, loss and BeamSearch n-best ------
    from espnet2.asr.decoder.rnn_decoder import RNNDecoder
    from espnet.nets.beam_search import BeamSearch as RefBeamSearch
    from espnet.nets.scorers.ctc import CTCPrefixScorer as RefCTCScorer
    from espnet.nets.scorers.length_bonus import LengthBonus as RefLengthBonus
    for tag, dkw in (("loc", dict(rnn_type="lstm", num_layers=2, att_conf=dict(atype="location", adim=8, aconv_chans=3,
                                                                              aconv_filts=4))),
                     ("mh", dict(rnn_type="gru", num_layers=1, context_residual=True,
                                 att_conf=dict(atype="multi_head_add", adim=8, aheads=2)))):
        torch.manual_seed(39)
        enc_m = RNNEncoder(20, num_layers=2, hidden_size=12, output_size=10, subsample=(2, 1))
        dec_m = RNNDecoder(30, 10, hidden_size=12, **dkw)
        m3 = ESPnetASRModel(vocab_size=30, token_list=["<blank>"] + [str(i) for i in range(1, 28)] + ["<space>", "<sos/eos>"],
                            frontend=None, specaug=None, normalize=None, encoder=enc_m, decoder=dec_m,
                            ctc=CTC2(30, 10, ctc_type="builtin"), rnnt_decoder=None, ctc_weight=0.3, lsm_weight=0.1,
                            length_normalized_loss=False)
        m3.train()
        sd3 = sd_np(m3, "sd/")
        loss3, stats3, w3 = m3(speech, slen, text, tlen)
        loss3.backward()
        m3.eval()
        res3 = {}
        with torch.no_grad():
            enc3, _ = m3.encode(speech[:1], slen[:1])
            for btag, cw in (("w00", 0.0), ("w03", 0.3)):
                scorers = dict(decoder=m3.decoder, ctc=RefCTCScorer(ctc=m3.ctc, eos=m3.eos), length_bonus=RefLengthBonus(30))
                bs = RefBeamSearch(beam_size=3, vocab_size=30, weights=dict(decoder=1.0 - cw, ctc=cw, length_bonus=0.1),
                                   scorers=scorers, sos=m3.sos, eos=m3.eos, token_list=None,
                                   pre_beam_score_key=None if cw == 1.0 else "full")
                nb = bs(x=enc3[0], maxlenratio=0.0, minlenratio=0.0)[:3]
                res3["beam_%s_scores" % btag] = np.asarray([float(h.score) for h in nb], dtype=np.float64)
                res3["beam_%s_lens" % btag] = np.asarray([len(h.yseq) for h in nb], dtype=np.int64)
                res3["beam_%s_yseq" % btag] = np.asarray(sum([[int(t) for t in h.yseq] for h in nb], []), dtype=np.int64)
        save(out("espnet2_rnn_%s.npz" % tag), speech=speech, speech_lengths=slen, text=text, text_lengths=tlen,
             loss=loss3.detach(), loss_att=stats3["loss_att"], loss_ctc=stats3["loss_ctc"], acc=float(stats3["acc"]), **res3,
             **sd3, **grads_np(m3))

    # ---- a19 / f2: BatchBeamSearch, and LM shallow fusion (TransformerLM, SequentialRNNLM) ----------
    from espnet.nets.batch_beam_search import BatchBeamSearch
    from espnet.nets.beam_search import BeamSearch as RefBeamSearch
    from espnet.nets.scorers.ctc import CTCPrefixScorer as RefCTCScorer
    from espnet.nets.scorers.length_bonus import LengthBonus as RefLengthBonus
    from espnet2.lm.seq_rnn_lm import SequentialRNNLM
    from espnet2.lm.transformer_lm import TransformerLM
    torch.manual_seed(29)
    tlm = TransformerLM(30, pos_enc=None, embed_unit=16, att_unit=32, head=2, unit=48, layer=2, dropout_rate=0.0)
    tlm_pe = TransformerLM(30, pos_enc="sinusoidal", embed_unit=16, att_unit=32, head=2, unit=48, layer=1,
                           dropout_rate=0.0)
    rlm = SequentialRNNLM(30, unit=24, nlayers=2, rnn_type="lstm")
    glm = SequentialRNNLM(30, unit=24, nhid=20, nlayers=1, rnn_type="gru")
    from espnet.nets.pytorch_backend.lm.default import DefaultRNNLM
    from espnet.nets.pytorch_backend.lm.transformer import TransformerLM as TransformerLM1
    dlm = DefaultRNNLM(30, argparse.Namespace(layer=2, unit=24, type="lstm", dropout_rate=0.0, embed_unit=None))
    dgm = DefaultRNNLM(30, argparse.Namespace(layer=1, unit=20, type="gru", dropout_rate=0.0, embed_unit=12))
    tlm1 = TransformerLM1(30, argparse.Namespace(layer=1, unit=40, att_unit=32, embed_unit=16, head=4,
                                                  dropout_rate=0.0, pos_enc="sinusoidal"))
    m2.eval()
    for lm_ in (tlm, tlm_pe, rlm, glm, dlm, dgm, tlm1):
        lm_.eval()
    fus = {}
    with torch.no_grad():
        toks = torch.randint(1, 29, (2, 6), generator=g)
        toks[1, 4:] = 0
        for nm, lm_ in (("tlm", tlm), ("tlm_pe", tlm_pe), ("rlm", rlm), ("glm", glm)):
            fus["lm_" + nm + "_logits"] = lm_(toks, None)[0]
        tgt = torch.cat([toks[:, 1:], torch.zeros(2, 1, dtype=toks.dtype)], dim=1)
        for nm, lm_ in (("dlm", dlm), ("dgm", dgm), ("tlm1", tlm1)):
            fus["lm_" + nm + "_loss"] = np.asarray([float(v) for v in lm_(toks, tgt)], dtype=np.float64)
        enc_out, _ = m2.encode(speech[:1], slen[:1])
        for tag, cls, lm_, cw, lw in (("bbeam_w00", BatchBeamSearch, None, 0.0, 0.0),
                                      ("bbeam_w03", BatchBeamSearch, None, 0.3, 0.0),
                                      ("bbeam_w10", BatchBeamSearch, None, 1.0, 0.0),
                                      ("bbeam_tlm", BatchBeamSearch, tlm, 0.3, 0.6),
                                      ("bbeam_tlm_pe", BatchBeamSearch, tlm_pe, 0.3, 0.6),
                                      ("bbeam_rlm", BatchBeamSearch, rlm, 0.3, 0.6),
                                      ("bbeam_glm", BatchBeamSearch, glm, 0.5, 0.4),
                                      ("bbeam_dlm", BatchBeamSearch, dlm, 0.3, 0.6),
                                      ("beam_dgm", RefBeamSearch, dgm, 0.3, 0.6),
                                      ("bbeam_tlm1", BatchBeamSearch, tlm1, 0.3, 0.6),
                                      ("beam_tlm", RefBeamSearch, tlm, 0.3, 0.6),
                                      ("beam_rlm", RefBeamSearch, rlm, 0.3, 0.6)):
            scorers = dict(decoder=m2.decoder, ctc=RefCTCScorer(ctc=m2.ctc, eos=m2.eos),
                           length_bonus=RefLengthBonus(30), lm=lm_)
            bs = cls(beam_size=4, vocab_size=30, weights=dict(decoder=1.0 - cw, ctc=cw, lm=lw, length_bonus=0.1),
                     scorers=scorers, sos=m2.sos, eos=m2.eos, token_list=None,
                     pre_beam_score_key=None if cw == 1.0 else "full")
            nb = bs(x=enc_out[0], maxlenratio=0.0, minlenratio=0.0)[:3]
            fus[tag + "_scores"] = np.asarray([float(h.score) for h in nb], dtype=np.float64)
            fus[tag + "_lens"] = np.asarray([len(h.yseq) for h in nb], dtype=np.int64)
            fus[tag + "_yseq"] = np.asarray(sum([[int(t) for t in h.yseq] for h in nb], []), dtype=np.int64)
            print(tag, fus[tag + "_scores"], [h.yseq.tolist() for h in nb][:1])
    save(out("decode_fusion.npz"), speech=speech[0], lm_tokens=toks, enc_out=enc_out[0], **fus, **sd_np(m2, "sd/"),
         **sd_np(tlm, "tlm/"), **sd_np(tlm_pe, "tlm_pe/"), **sd_np(rlm, "rlm/"), **sd_np(glm, "glm/"),
         **sd_np(dlm, "dlm/"), **sd_np(dgm, "dgm/"), **sd_np(tlm1, "tlm1/"))

    # ---- a20: RNN path (VGG-BLSTMP encoder, location-aware attention LSTM decoder, CTC) -----------
    from espnet.nets.pytorch_backend.e2e_asr import E2E as RnnE2E

    def rnn_args(**kw):
        d = dict(elayers=2, subsample="1_2_1", etype="vggblstmp", eunits=12, eprojs=10, dtype="lstm", dlayers=2,
                 dunits=14, atype="location", aheads=1, awin=3, aconv_chans=3, aconv_filts=2, mtlalpha=0.5,
                 lsm_type="", lsm_weight=0.0, sampling_probability=0.0, adim=9, dropout_rate=0.0,
                 dropout_rate_decoder=0.0, nbest=1, beam_size=1, penalty=0.0, maxlenratio=0.0, minlenratio=0.0,
                 ctc_weight=0.0, ctc_window_margin=0, lm_weight=0.0, rnnlm=None, verbose=0,
                 char_list=["<blank>", "a", "b", "c", "d", "e", "<eos>"], outdir=None, ctc_type="builtin",
                 report_cer=False, report_wer=False, sym_space="<space>", sym_blank="<blank>", sortagrad=0,
                 grad_noise=False, context_residual=False, use_frontend=False, replace_sos=False, tgt_lang=False)
        d.update(kw)
        return argparse.Namespace(**d)

    def rnn_decode(model, x, seed):
        """a20 decode: E2E.recognize -> Decoder.recognize_beam (rnn/decoders.py:313-605) with and without CTC / LM"""
        from espnet.nets.pytorch_backend.lm.default import ClassifierWithState, RNNLM
        torch.manual_seed(seed)
        lm = ClassifierWithState(RNNLM(7, 1, 8, None, "lstm", 0.0)).eval()
        res = dict(sd_np(lm, "rlm/"))
        was = model.training
        model.eval()
        with torch.no_grad():
            for tag, kw, use_lm in (("b3", dict(beam_size=3, ctc_weight=0.0, penalty=0.0), False),
                                    ("b3ctc", dict(beam_size=3, ctc_weight=0.5, penalty=0.1), False),
                                    ("b2lm", dict(beam_size=2, ctc_weight=0.3, penalty=0.0, lm_weight=0.4), True),
                                    ("b3ctc1", dict(beam_size=3, ctc_weight=1.0, penalty=0.2, maxlenratio=0.5), False)):
                ra = argparse.Namespace(**dict(dict(nbest=3, maxlenratio=0.0, minlenratio=0.0, lm_weight=0.0), **kw))
                nb = model.recognize(x, ra, model.char_list if hasattr(model, "char_list") else rnn_args().char_list,
                                     lm if use_lm else None)
                res["rb_%s_scores" % tag] = np.asarray([float(h["score"]) for h in nb], dtype=np.float64)
                res["rb_%s_lens" % tag] = np.asarray([len(h["yseq"]) for h in nb], dtype=np.int64)
                res["rb_%s_yseq" % tag] = np.asarray(sum([[int(t) for t in h["yseq"]] for h in nb], []), dtype=np.int64)
        model.train(was)
        return res

    torch.manual_seed(31)
    m = RnnE2E(12, 7, rnn_args())
    m.train()
    sd0 = sd_np(m, "sd/")
    g = torch.Generator().manual_seed(3)
    xs = torch.randn(3, 41, 12, generator=g)
    ilens = torch.tensor([41, 33, 20])
    ys = torch.randint(1, 6, (3, 6), generator=g)
    ys[1, 4:] = -1
    ys[2, 3:] = -1
    xs = xs * (torch.arange(41).view(1, -1, 1) < ilens.view(-1, 1, 1))
    hs, hlens, _ = m.enc(xs, ilens)
    loss = m(xs, ilens, ys)
    loss.backward()
    save(out("e2e_rnn.npz"), xs=xs, ilens=ilens, ys=ys, hs_pad=hs.detach(), hlens=np.asarray(hlens, dtype=np.int64),
         loss=float(loss), loss_att=float(m.loss_att), loss_ctc=float(m.loss_ctc), acc=float(m.acc), **sd0,
         **grads_np(m), **rnn_decode(m, xs[0].numpy(), 131))

    # a20: the other attention types on the HIP path, on a small BLSTMP (subsampling 1_2) model each
    for atype in ("dot", "add", "multi_head_dot", "multi_head_add", "multi_head_loc", "multi_head_multi_res_loc", "noatt",
                  "coverage", "coverage_location", "location2d", "location_recurrent"):
        torch.manual_seed(32)
        m = RnnE2E(9, 7, rnn_args(etype="blstmp", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dlayers=1, dunits=10,
                                  atype=atype, adim=6, aheads=2, aconv_chans=3, aconv_filts=4))
        m.train()
        sd0 = sd_np(m, "sd/")
        g = torch.Generator().manual_seed(5)
        xs = torch.randn(3, 26, 9, generator=g)
        ilens = torch.tensor([26, 21, 14])
        ys = torch.randint(1, 6, (3, 5), generator=g)
        ys[1, 3:] = -1
        ys[2, 4:] = -1
        xs = xs * (torch.arange(26).view(1, -1, 1) < ilens.view(-1, 1, 1))
        hs, hlens, _ = m.enc(xs, ilens)
        loss = m(xs, ilens, ys)
        loss.backward()
        save(out("e2e_rnn_%s.npz" % atype), xs=xs, ilens=ilens, ys=ys, hs_pad=hs.detach(),
             hlens=np.asarray(hlens, dtype=np.int64), loss=float(loss), loss_att=float(m.loss_att),
             loss_ctc=float(m.loss_ctc), acc=float(m.acc), **sd0, **grads_np(m), **rnn_decode(m, xs[0].numpy(), 132))

    # a20: GRU cells (bidirectional GRU-P encoder, 2-layer GRU decoder)
    torch.manual_seed(33)
    m = RnnE2E(9, 7, rnn_args(etype="bgrup", elayers=2, subsample="1_2_1", eunits=8, eprojs=8, dtype="gru", dlayers=2,
                              dunits=10, atype="location", adim=6, aconv_chans=3, aconv_filts=4))
    m.train()
    sd0 = sd_np(m, "sd/")
    g = torch.Generator().manual_seed(6)
    xs = torch.randn(3, 26, 9, generator=g)
    ilens = torch.tensor([26, 21, 14])
    ys = torch.randint(1, 6, (3, 5), generator=g)
    ys[1, 3:] = -1
    xs = xs * (torch.arange(26).view(1, -1, 1) < ilens.view(-1, 1, 1))
    hs, hlens, _ = m.enc(xs, ilens)
    loss = m(xs, ilens, ys)
    loss.backward()
    save(out("e2e_rnn_gru.npz"), xs=xs, ilens=ilens, ys=ys, hs_pad=hs.detach(), hlens=np.asarray(hlens, dtype=np.int64),
         loss=float(loss), loss_att=float(m.loss_att), loss_ctc=float(m.loss_ctc), acc=float(m.acc), **sd0, **grads_np(m),
         **rnn_decode(m, xs[0].numpy(), 133))

    # ---- a21: transducer.  The loss package (warprnnt_pytorch) is absent here: TransLoss is given the
    # oracle's float64 restatement of the published recursion (asr_oracle.rnnt_loss, mean over the batch),
    # everything else (encoder, predictor, joint network, input preparation) is the reference's own code.
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import asr_oracle

    class _RNNTLoss:
        def __init__(self, blank=0, **kw):
            self.blank = blank

        def __call__(self, acts, labels, act_lens, label_lens):
            return asr_oracle.rnnt_loss(acts, labels, act_lens, label_lens, self.blank).to(acts.dtype)

    wp = types.ModuleType("warprnnt_pytorch")
    wp.RNNTLoss = _RNNTLoss
    sys.modules["warprnnt_pytorch"] = wp
    from espnet.nets.pytorch_backend.e2e_asr_transducer import E2E as TrnE2E

    def trn_case(name, seed, **kw):
        d = dict(etype="vggblstmp", elayers=1, subsample="1_1", eunits=10, eprojs=8, dtype="lstm", dlayers=2,
                 dunits=12, dec_embed_dim=6, atype="location", adim=4, aheads=2, awin=2, aconv_chans=2,
                 aconv_filts=5, dropout_rate=0.0, dropout_rate_decoder=0.0, dropout_rate_embed_decoder=0.0,
                 joint_dim=7, joint_activation_type="tanh", mtlalpha=1.0, rnnt_mode="rnnt", use_frontend=False,
                 trans_type="warp-transducer", char_list=["a", "b", "c", "d"], sym_space="<space>",
                 sym_blank="<blank>", report_cer=False, report_wer=False, score_norm_transducer=True, beam_size=1,
                 nbest=1, verbose=0, outdir=None, rnnlm=None, transformer_init="pytorch")
        d.update(kw)
        torch.manual_seed(seed)
        m = TrnE2E(12, 6, argparse.Namespace(**d))
        m.train()
        sd0 = sd_np(m, "sd/")
        g = torch.Generator().manual_seed(seed + 100)
        xs = torch.randn(3, 37, 12, generator=g)
        ilens = torch.tensor([37, 30, 21])
        ys = torch.randint(1, 6, (3, 5), generator=g)
        ys[1, 3:] = -1
        ys[2, 4:] = -1
        xs = xs * (torch.arange(37).view(1, -1, 1) < ilens.view(-1, 1, 1))
        loss = m(xs, ilens, ys)
        loss.backward()
        # decoding (beam_search_transducer.py:130-237): greedy and default beam search on the first utterance
        from espnet.nets.beam_search_transducer import BeamSearchTransducer
        hs_train, pred_train = m.hs_pad.detach().clone(), m.pred_pad.detach().clone()
        dec = {}
        m.eval()
        with torch.no_grad():
            xin = xs[0, : int(ilens[0])].numpy()
            for tag, kw2 in (("greedy", dict(beam_size=1)), ("beam3", dict(beam_size=3, search_type="default")),
                             ("beam3_nonorm", dict(beam_size=3, search_type="default", score_norm=False)),
                             ("tsd3", dict(beam_size=3, search_type="tsd", max_sym_exp=2)),
                             ("tsd2", dict(beam_size=2, search_type="tsd", max_sym_exp=3, score_norm=False)),
                             ("alsd3", dict(beam_size=3, search_type="alsd", u_max=10)),
                             ("alsd2", dict(beam_size=2, search_type="alsd", u_max=4, score_norm=False)),
                             ("nsc3", dict(beam_size=3, search_type="nsc", nstep=1, prefix_alpha=1)),
                             ("nsc3n2", dict(beam_size=3, search_type="nsc", nstep=2, prefix_alpha=2)),
                             ("nsc2n3", dict(beam_size=2, search_type="nsc", nstep=3, prefix_alpha=1, score_norm=False))):
                if d["rnnt_mode"] == "rnnt-att" and kw2.get("search_type", "default") != "default":
                    continue      # the batched searches are not usable with the attention decoder's batch_score
                bs = BeamSearchTransducer(decoder=m.decoder if hasattr(m, "decoder") else m.dec, lm=None, lm_weight=0.0, **kw2)
                nb = m.recognize(xin, bs)
                nb = nb if isinstance(nb, list) else [nb]
                dec["dec_%s_scores" % tag] = np.asarray([float(h["score"]) for h in nb], dtype=np.float64)
                dec["dec_%s_lens" % tag] = np.asarray([len(h["yseq"]) for h in nb], dtype=np.int64)
                dec["dec_%s_yseq" % tag] = np.asarray(sum([[int(t) for t in h["yseq"]] for h in nb], []), dtype=np.int64)
            # LM shallow fusion in the default search (beam_search_transducer.py:204-224): an espnet1 RNNLM
            from espnet.nets.pytorch_backend.lm.default import ClassifierWithState, RNNLM
            torch.manual_seed(seed + 7)
            lm = ClassifierWithState(RNNLM(6, 1, 8, None, "lstm", 0.0)).eval()
            for tag, st in (("beam3_lm", "default"), ("tsd3_lm", "tsd"), ("alsd3_lm", "alsd"), ("nsc3_lm", "nsc")):
                if d["rnnt_mode"] == "rnnt-att" and st != "default":
                    continue
                bs = BeamSearchTransducer(decoder=m.decoder if hasattr(m, "decoder") else m.dec, lm=lm, lm_weight=0.5, beam_size=3,
                                          search_type=st, nstep=2)
                nb = m.recognize(xin, bs)
                dec["dec_%s_scores" % tag] = np.asarray([float(h["score"]) for h in nb], dtype=np.float64)
                dec["dec_%s_lens" % tag] = np.asarray([len(h["yseq"]) for h in nb], dtype=np.int64)
                dec["dec_%s_yseq" % tag] = np.asarray(sum([[int(t) for t in h["yseq"]] for h in nb], []), dtype=np.int64)
            dec.update(sd_np(lm, "lm/"))
        save(out(name), xs=xs, ilens=ilens, ys=ys, hs_pad=hs_train, pred_pad=pred_train,
             loss=float(loss), **dec, **sd0, **grads_np(m))

    trn_case("transducer_rnn.npz", 41)
    trn_case("transducer_gru.npz", 43, etype="bgru", elayers=2, dtype="gru", dlayers=2)   # stacked nn.GRU + l_last
    conf_arch = [dict(type="conformer", d_hidden=64, d_ff=96, heads=4, macaron_style=True, use_conv_mod=True,
                      conv_mod_kernel=7)]
    conf_arch[0]["dropout-rate"] = 0.0
    conf_arch[0]["pos-dropout-rate"] = 0.0
    conf_arch[0]["att-dropout-rate"] = 0.0
    # transformer-transducer: transformer prediction network (DecoderTT) behind a transformer encoder (the reference's
    # initializer breaks on an RNN encoder + transformer decoder: transducer/initializer.py:34 reads model.encoder)
    tt_dec = [dict(type="transformer", d_hidden=16, d_ff=24, heads=2)]
    tt_kw = dict(dtype="transformer", dec_block_arch=tt_dec, dec_block_repeat=2, transformer_dec_input_layer="embed",
                 transformer_dec_pw_activation_type="relu")
    trn_case("transducer_tt.npz", 45, etype="transformer",
             enc_block_arch=[dict(type="transformer", d_hidden=64, d_ff=96, heads=4)], enc_block_repeat=2,
             transformer_enc_input_layer="conv2d", transformer_enc_self_attn_type="self_attn",
             transformer_enc_positional_encoding_type="abs_pos", transformer_enc_pw_activation_type="relu",
             transformer_enc_conv_mod_activation_type="relu", **tt_kw)
    trn_case("transducer_att.npz", 46, rnnt_mode="rnnt-att", etype="blstmp", elayers=1, subsample="1_1", atype="location")
    trn_case("transducer_att_gru.npz", 47, rnnt_mode="rnnt-att", etype="blstmp", elayers=1, subsample="1_1", dtype="gru",
             dlayers=1, atype="multi_head_add")
    trn_case("transducer_conformer.npz", 42, etype="transformer", enc_block_arch=conf_arch, enc_block_repeat=2,
             transformer_enc_input_layer="conv2d", transformer_enc_self_attn_type="rel_self_attn",
             transformer_enc_positional_encoding_type="rel_pos", transformer_enc_pw_activation_type="swish",
             transformer_enc_conv_mod_activation_type="swish", dlayers=1, joint_activation_type="tanh")

    # ---- 8f rank 1: SpecAug and the normalisation layers (seeded draws from the CPU generator) -------------
    from espnet2.asr.specaug.specaug import SpecAug
    from espnet2.layers.global_mvn import GlobalMVN
    from espnet2.layers.utterance_mvn import UtteranceMVN
    import tempfile
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(4, 120, 20, generator=g) * 2.0 + 0.5
    res = {}
    for tag, lens in (("eq", [120, 120, 120, 120]), ("ragged", [120, 97, 64, 9])):
        ll = torch.tensor(lens)
        x = feats * (torch.arange(120).view(1, -1, 1) < ll.view(-1, 1, 1))
        sa = SpecAug(time_warp_window=5, freq_mask_width_range=(0, 6), num_freq_mask=2, time_mask_width_range=(0, 20),
                     num_time_mask=2)
        torch.manual_seed(77)
        y, _ = sa(x.clone(), ll)
        res["specaug_%s" % tag] = y
        res["lens_%s" % tag] = ll
        sa2 = SpecAug(apply_time_warp=False, freq_mask_width_range=(0, 6), time_mask_width_range=(0, 20))
        torch.manual_seed(78)
        res["specaug_nowarp_%s" % tag] = sa2(x.clone(), ll)[0]
        sa3 = SpecAug(apply_freq_mask=False, apply_time_mask=False, time_warp_window=7)
        torch.manual_seed(79)
        res["specaug_warponly_%s" % tag] = sa3(x.clone(), ll)[0]
    cnt = 1000.0
    ssum = (torch.randn(20, generator=g) * cnt).double().numpy()
    ssq = (torch.rand(20, generator=g).double().numpy() * 4.0 + 1.0) * cnt + ssum * ssum / cnt
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "stats.npz"), count=cnt, sum=ssum, sum_square=ssq)
        ll = torch.tensor([120, 97, 64, 9])
        x = feats * (torch.arange(120).view(1, -1, 1) < ll.view(-1, 1, 1))
        for nm in (True, False):
            for nv in (True, False):
                gm = GlobalMVN(os.path.join(td, "stats.npz"), norm_means=nm, norm_vars=nv)
                res["gmvn_%d%d" % (nm, nv)] = gm(x.clone(), ll)[0]
                um = UtteranceMVN(norm_means=nm, norm_vars=nv)
                res["umvn_%d%d" % (nm, nv)] = um(x.clone(), ll)[0]
    save(out("feature_layers.npz"), feats=feats, stats_count=cnt, stats_sum=ssum, stats_sum_square=ssq, **res)

    # ---- 8f rank 4: log-mel frontend.  The reference's Stft calls torch.stft without `return_complex`, which the
    # installed torch requires; the call below is the same one with that flag (and view_as_real for the (..., 2)
    # layout stft.py:92-95 expects).  librosa is absent: LogMel's `librosa.filters.mel` resolves to the oracle's
    # restatement of that function (a stub module, like warprnnt_pytorch above).
    import asr_oracle as _orc
    lib = types.ModuleType("librosa")
    lib.filters = types.ModuleType("librosa.filters")
    lib.filters.mel = lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False: _orc.mel_filterbank(
        sr, n_fft, n_mels, fmin, fmax, htk)
    sys.modules["librosa"], sys.modules["librosa.filters"] = lib, lib.filters
    from espnet2.layers.log_mel import LogMel
    from espnet2.layers.stft import Stft
    _stft = torch.stft
    torch.stft = lambda *a, **k: torch.view_as_real(_stft(*a, return_complex=True, **k))
    try:
        g = torch.Generator().manual_seed(13)
        wav = torch.randn(3, 4000, generator=g) * torch.linspace(0.05, 1.0, 4000)
        wav = wav + 0.3 * torch.sin(torch.arange(4000) * 0.05)
        wlens = torch.tensor([4000, 3301, 1500])
        wav = wav * (torch.arange(4000)[None, :] < wlens[:, None])
        fr = {}
        for tag, kw_s, kw_m in (("default", dict(), dict()),
                                ("win400", dict(n_fft=512, win_length=400, hop_length=160), dict(n_mels=40, htk=True)),
                                ("n256", dict(n_fft=256, hop_length=64), dict(n_fft=256, n_mels=23, fmin=80, fmax=7600))):
            st = Stft(**kw_s)
            spec, flens = st(wav, wlens)
            power = spec[..., 0] ** 2 + spec[..., 1] ** 2          # frontend/default.py:121-124
            feats, _ = LogMel(**kw_m)(power, flens)
            fr[tag + "_stft"], fr[tag + "_flens"], fr[tag + "_feats"] = spec, flens, feats
        wav2 = torch.stack([wav, wav.flip(0)], dim=-1)             # (B, L, C): channel 0 is used in eval mode
        fr["mc_stft"] = Stft()(wav2, wlens)[0]
    finally:
        torch.stft = _stft
    save(out("frontend.npz"), wav=wav, wlens=wlens, **fr)

    e2e_case("e2e_conformer.npz", ConfE2E, dict(transformer_encoder_pos_enc_layer_type="rel_pos",
                                                transformer_encoder_selfattn_layer_type="rel_selfattn",
                                                transformer_encoder_activation_type="swish", macaron_style=True,
                                                use_cnn_module=True, cnn_module_kernel=15), 21)
    e2e_case("e2e_transformer.npz", TrfE2E, dict(eunits=256, dunits=256), 22)


if __name__ == "__main__":
    main()
