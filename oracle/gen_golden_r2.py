#!/usr/bin/env python3
"""Round-2 fixtures, again by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU) in the build container:

  rel_mha_dk64.npz        RelPositionMultiHeadedAttention(4, 256): d_k = 64, T = 249 / ragged mask + a dead utterance
  mha_dk64.npz            MultiHeadedAttention(4, 256): causal self-attention (T = 101) and source attention (101 x 249)
  e2e_conformer_dk64.npz  espnet1 Conformer E2E with adim 256 / aheads 4 (the width at which bf16 mode dispatches the
                          fused attention kernels), 2 encoder layers, 1 decoder layer: loss, loss_ctc, acc, hs_pad,
                          greedy ids, every parameter gradient (whole or as two random projections)
  scaled_posenc.npz       ScaledPositionalEncoding (embedding.py:95-128) forward / backward incl. d alpha

Weights are NOT stored: both sides fill them from oracle/seeded_weights.py (name-keyed generator).
Usage: python oracle/gen_golden_r2.py [--ref /root/reference] [--out tests/golden]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    install_stubs()
    sys.path.insert(0, a.ref)
    out = lambda n: os.path.join(a.out, n)  # noqa: E731
    torch.set_num_threads(4)

    from espnet.nets.pytorch_backend.transformer.attention import (MultiHeadedAttention,
                                                                    RelPositionMultiHeadedAttention)
    from espnet.nets.pytorch_backend.transformer.embedding import (RelPositionalEncoding, ScaledPositionalEncoding)
    from espnet.nets.pytorch_backend.transformer.mask import subsequent_mask

    def grads(module):
        rec = {}
        for name, p in module.named_parameters():
            if p.grad is not None:
                rec.update(SW.grad_record(name, p.grad))
        return rec

    # ---- a7 at the dispatched width: d = 256, h = 4 (d_k = 64), T' = 249 as at config 2 ----
    att = SW.fill_parameters(RelPositionMultiHeadedAttention(4, 256, 0.0), salt=71)
    g = torch.Generator().manual_seed(71)
    B, T = 3, 249
    x = torch.randn(B, T, 256, generator=g).requires_grad_(True)
    _, pos = RelPositionalEncoding(256, 0.0)(x.detach())
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, 170:] = False
    mask[2, 0, :] = False                   # a fully masked utterance: attention.py:84-88 gives zeros
    y = att(x, x, x, pos, mask)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    save(out("rel_mha_dk64.npz"), x=x.detach(), pos=pos.detach(), mask=mask, y=y.detach(), gy=gy, gx=x.grad,
         attn_sample=att.attn.detach()[:, :, ::31, :], **grads(att))

    # ---- a6 at d_k = 64: causal self-attention and source attention over a longer memory ----
    att = SW.fill_parameters(MultiHeadedAttention(4, 256, 0.0), salt=61)
    g = torch.Generator().manual_seed(61)
    q = torch.randn(2, 101, 256, generator=g).requires_grad_(True)
    mem = torch.randn(2, 249, 256, generator=g).requires_grad_(True)
    mmask = torch.ones(2, 1, 249, dtype=torch.bool)
    mmask[1, 0, 150:] = False
    y = att(q, mem, mem, mmask)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    rec_src = grads(att)
    gq, gmem = q.grad.clone(), mem.grad.clone()
    att.zero_grad()
    q2 = q.detach().clone().requires_grad_(True)
    cm = subsequent_mask(101).unsqueeze(0).expand(2, 101, 101).clone()
    cm[1, :, 80:] = False
    y2 = att(q2, q2, q2, cm)
    y2.backward(gy)
    rec_self = {"self_" + k: v for k, v in grads(att).items()}
    save(out("mha_dk64.npz"), q=q.detach(), mem=mem.detach(), mmask=mmask, y=y.detach(), gy=gy, gq=gq, gmem=gmem,
         cmask=cm, y_self=y2.detach(), gq_self=q2.grad, **rec_src, **rec_self)

    # ---- a4 ScaledPositionalEncoding ----
    pe = ScaledPositionalEncoding(64, 0.0)
    with torch.no_grad():
        pe.alpha.fill_(0.7)
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 13, 64, generator=g).requires_grad_(True)
    y = pe(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    save(out("scaled_posenc.npz"), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, galpha=pe.alpha.grad, alpha=pe.alpha.detach())

    # ---- a1 at the dispatched width: espnet1 Conformer E2E, adim 256, aheads 4 ----
    from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E as ConfE2E
    ns = argparse.Namespace(
        adim=256, aheads=4, elayers=2, eunits=64, dlayers=1, dunits=64, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=0.0,
        transformer_attn_dropout_rate=0.0, transformer_length_normalized_loss=False, transformer_init="pytorch",
        transformer_input_layer="conv2d", ctc_type="builtin", report_cer=False, report_wer=False, char_list=None,
        sym_space="<space>", sym_blank="<blank>", transformer_encoder_pos_enc_layer_type="rel_pos",
        transformer_encoder_selfattn_layer_type="rel_selfattn", transformer_encoder_activation_type="swish",
        macaron_style=True, use_cnn_module=True, cnn_module_kernel=31)
    torch.manual_seed(5)
    model = SW.fill_parameters(ConfE2E(20, 50, ns), salt=5)
    model.train()
    g = torch.Generator().manual_seed(5)
    xs = torch.randn(3, 300, 20, generator=g)
    ilens = torch.tensor([300, 251, 180])
    ys = torch.randint(1, 49, (3, 12), generator=g)
    ys[1, 9:] = -1
    ys[2, 5:] = -1
    loss = model(xs, ilens, ys)
    loss.backward()
    rec = dict(loss=float(loss), acc=float(model.acc), hs_pad=model.hs_pad.detach().clone(),
               pred_pad=model.pred_pad.detach().clone(), loss_ctc=float(model.ctc.loss))
    rec.update(grads(model))
    model.eval()
    with torch.no_grad():
        from itertools import groupby
        greedy, glens = [], []
        for b in range(3):
            lz = model.ctc.argmax(model.encoder(xs[b:b + 1, :int(ilens[b])], None)[0])
            ids = [v for v in (k[0] for k in groupby(lz[0].tolist())) if v != 0]
            greedy += ids
            glens.append(len(ids))
    save(out("e2e_conformer_dk64.npz"), xs=xs, ilens=ilens, ys=ys, greedy=np.asarray(greedy, dtype=np.int64),
         greedy_lens=np.asarray(glens, dtype=np.int64), **rec)


if __name__ == "__main__":
    main()
