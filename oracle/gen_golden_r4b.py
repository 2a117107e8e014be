#!/usr/bin/env python3
"""Round-4 fixture (b), by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU) in the build container:

  postnorm_layers.npz     the layers with normalize_before=False and / or concat_after=True (conformer/encoder_layer.py:99-157,
                          transformer/encoder_layer.py:53-101, transformer/decoder_layer.py:60-134; size 64, 4 heads, units 96,
                          macaron + convolution module kernel 7): four variants (post-norm, post-norm + concat, pre-norm + concat)
                          of each of the three layers - outputs, input gradients, every parameter gradient; the decoder layer
                          also through its cached form (the newest position only)
  e2e_conformer_long.npz  the model of e2e_conformer_dk64.npz (adim 256 / aheads 4, seeded weights salt 5) on LONG inputs: two
                          utterances of 2200 / 1777 frames (T' = 549 / 443: attention rows beyond 512 keys, the legacy rel_shift on
                          a padded batch), 30 / 22 labels - loss, loss_ctc, acc, hs_pad, every parameter gradient
  e2e_conformer_d512.npz  espnet1 Conformer E2E at the width of the reference's large recipes (egs/librispeech/asr1 conformer:
                          adim 512, aheads 8 - d_k = 64 -, eunits = dunits = 2048), 2 encoder layers, 1 decoder layer,
                          idim 80, |V| = 50, three utterances of 300 / 251 / 180 frames: loss, loss_ctc, acc, hs_pad, pred_pad,
                          greedy CTC ids, every parameter gradient (whole where small, else as two random projections:
                          oracle/seeded_weights.py grad_record).

Weights are NOT stored: both sides fill them from oracle/seeded_weights.py (name-keyed generator, salt 512).
Usage: python oracle/gen_golden_r4b.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import install_stubs, save  # noqa: E402
import seeded_weights as SW  # noqa: E402

D512 = dict(idim=80, odim=50, salt=512, seed=512, lens=(300, 251, 180),
            ns=dict(adim=512, aheads=8, elayers=2, eunits=2048, dlayers=1, dunits=2048, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=0.0,
                    transformer_attn_dropout_rate=0.0, transformer_length_normalized_loss=False, transformer_init="pytorch",
                    transformer_input_layer="conv2d", ctc_type="builtin", report_cer=False, report_wer=False, char_list=None,
                    sym_space="<space>", sym_blank="<blank>", transformer_encoder_pos_enc_layer_type="rel_pos",
                    transformer_encoder_selfattn_layer_type="rel_selfattn", transformer_encoder_activation_type="swish",
                    macaron_style=True, use_cnn_module=True, cnn_module_kernel=31))


def d512_inputs():
    g = torch.Generator().manual_seed(D512["seed"])
    xs = torch.randn(3, 300, D512["idim"], generator=g)
    ilens = torch.tensor(D512["lens"])
    ys = torch.randint(1, D512["odim"] - 1, (3, 12), generator=g)
    ys[1, 9:] = -1
    ys[2, 5:] = -1
    return xs, ilens, ys


POSTNORM_VARIANTS = (("post", False, False), ("postcat", False, True), ("precat", True, True))


def postnorm_layers(out_path):
    from espnet.nets.pytorch_backend.conformer.convolution import ConvolutionModule
    from espnet.nets.pytorch_backend.conformer.encoder_layer import EncoderLayer as ConfLayer
    from espnet.nets.pytorch_backend.conformer.swish import Swish
    from espnet.nets.pytorch_backend.transformer.attention import MultiHeadedAttention, RelPositionMultiHeadedAttention
    from espnet.nets.pytorch_backend.transformer.decoder_layer import DecoderLayer
    from espnet.nets.pytorch_backend.transformer.embedding import RelPositionalEncoding
    from espnet.nets.pytorch_backend.transformer.encoder_layer import EncoderLayer as TrfLayer
    from espnet.nets.pytorch_backend.transformer.mask import subsequent_mask
    from espnet.nets.pytorch_backend.transformer.positionwise_feed_forward import PositionwiseFeedForward
    D, H, U = 64, 4, 96
    g = torch.Generator().manual_seed(99)
    B, T, L = 3, 37, 11
    x = torch.randn(B, T, D, generator=g)
    _, pos = RelPositionalEncoding(D, 0.0)(x)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, 30:] = False
    tgt = torch.randn(B, L, D, generator=g)
    tmask = subsequent_mask(L).unsqueeze(0).expand(B, L, L).clone()
    gy, gyt = torch.randn(B, T, D, generator=g), torch.randn(B, L, D, generator=g)
    rec = dict(x=x, pos=pos, mask=mask, tgt=tgt, tmask=tmask, gy=gy, gyt=gyt)

    def grads(tag, module):
        for name, p in module.named_parameters():
            if p.grad is not None:
                rec["%s/grad/%s" % (tag, name)] = p.grad.detach().clone()

    for tag, nb, cat in POSTNORM_VARIANTS:
        conf = ConfLayer(D, RelPositionMultiHeadedAttention(H, D, 0.0), PositionwiseFeedForward(D, U, 0.0, Swish()),
                         PositionwiseFeedForward(D, U, 0.0, Swish()), ConvolutionModule(D, 7, Swish()), 0.0, nb, cat)
        conf = SW.fill_parameters(conf, salt=990).train()
        xi = x.clone().requires_grad_(True)
        (y, _), _ = conf((xi, pos), mask)
        y.backward(gy)
        rec["conf_%s/y" % tag], rec["conf_%s/gx" % tag] = y.detach().clone(), xi.grad.clone()
        grads("conf_" + tag, conf)
        trf = SW.fill_parameters(TrfLayer(D, MultiHeadedAttention(H, D, 0.0), PositionwiseFeedForward(D, U, 0.0), 0.0, nb, cat), salt=991).train()
        xi = x.clone().requires_grad_(True)
        y, _ = trf(xi, mask)
        y.backward(gy)
        rec["trf_%s/y" % tag], rec["trf_%s/gx" % tag] = y.detach().clone(), xi.grad.clone()
        grads("trf_" + tag, trf)
        dec = SW.fill_parameters(DecoderLayer(D, MultiHeadedAttention(H, D, 0.0), MultiHeadedAttention(H, D, 0.0),
                                              PositionwiseFeedForward(D, U, 0.0), 0.0, nb, cat), salt=992).train()
        ti, mi = tgt.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y, *_ = dec(ti, tmask, mi, mask)
        y.backward(gyt)
        rec["dec_%s/y" % tag], rec["dec_%s/gtgt" % tag], rec["dec_%s/gmem" % tag] = y.detach().clone(), ti.grad.clone(), mi.grad.clone()
        grads("dec_" + tag, dec)
        with torch.no_grad():        # the cached form: prefix outputs of positions < L - 1 given, the newest position computed
            yc, *_ = dec.eval()(tgt, tmask, x, mask, cache=y.detach()[:, :-1])
        rec["dec_%s/y_cached" % tag] = yc.clone()
    save(out_path, **rec)


def long_inputs(out_path, E2E):
    ns = argparse.Namespace(
        adim=256, aheads=4, elayers=2, eunits=64, dlayers=1, dunits=64, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=0.0,
        transformer_attn_dropout_rate=0.0, transformer_length_normalized_loss=False, transformer_init="pytorch",
        transformer_input_layer="conv2d", ctc_type="builtin", report_cer=False, report_wer=False, char_list=None,
        sym_space="<space>", sym_blank="<blank>", transformer_encoder_pos_enc_layer_type="rel_pos",
        transformer_encoder_selfattn_layer_type="rel_selfattn", transformer_encoder_activation_type="swish",
        macaron_style=True, use_cnn_module=True, cnn_module_kernel=31)
    torch.manual_seed(5)
    model = SW.fill_parameters(E2E(20, 50, ns), salt=5)
    model.train()
    g = torch.Generator().manual_seed(2200)
    xs = torch.randn(2, 2200, 20, generator=g)
    ilens = torch.tensor([2200, 1777])
    ys = torch.randint(1, 49, (2, 30), generator=g)
    ys[1, 22:] = -1
    loss = model(xs, ilens, ys)
    loss.backward()
    rec = dict(loss=float(loss), acc=float(model.acc), hs_pad=model.hs_pad.detach()[:, ::4].clone(), loss_ctc=float(model.ctc.loss))
    for name, p in model.named_parameters():
        if p.grad is not None:
            rec.update(SW.grad_record(name, p.grad))
    # inputs are regenerated by the test from the same generator (2 x 2200 x 20 floats are not stored)
    save(out_path, ilens=ilens, ys=ys, **rec)
    print("long: loss %.6f ctc %.6f acc %.4f" % (float(loss), float(model.ctc.loss), float(model.acc)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    install_stubs()
    sys.path.insert(0, a.ref)
    torch.set_num_threads(8)
    from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E
    postnorm_layers(os.path.join(a.out, "postnorm_layers.npz"))
    long_inputs(os.path.join(a.out, "e2e_conformer_long.npz"), E2E)
    torch.manual_seed(D512["seed"])
    model = SW.fill_parameters(E2E(D512["idim"], D512["odim"], argparse.Namespace(**D512["ns"])), salt=D512["salt"])
    model.train()
    xs, ilens, ys = d512_inputs()
    loss = model(xs, ilens, ys)
    loss.backward()
    rec = dict(loss=float(loss), acc=float(model.acc), hs_pad=model.hs_pad.detach().clone(), pred_pad=model.pred_pad.detach().clone(),
               loss_ctc=float(model.ctc.loss))
    for name, p in model.named_parameters():
        if p.grad is not None:
            rec.update(SW.grad_record(name, p.grad))
    model.eval()
    with torch.no_grad():
        from itertools import groupby
        greedy, glens = [], []
        for b in range(3):
            lz = model.ctc.argmax(model.encoder(xs[b:b + 1, :int(ilens[b])], None)[0])
            ids = [v for v in (k[0] for k in groupby(lz[0].tolist())) if v != 0]
            greedy += ids
            glens.append(len(ids))
    save(os.path.join(a.out, "e2e_conformer_d512.npz"), xs=xs, ilens=ilens, ys=ys, greedy=np.asarray(greedy, dtype=np.int64),
         greedy_lens=np.asarray(glens, dtype=np.int64), **rec)
    print("loss %.6f ctc %.6f acc %.4f greedy lens %s" % (float(loss), float(model.ctc.loss), float(model.acc), glens))


if __name__ == "__main__":
    main()
