#!/usr/bin/env python3
"""Round-3 fixtures, again by RUNNING THE REFERENCE (kan-bayashi/espnet v0.9.5, PyTorch CPU) in the build container
(test infrastructure: nothing here is imported by the product):

  ffn_{hardtanh,tanh,selu}.npz   PositionwiseFeedForward with the other three activations of nets_utils.get_activation
                                 (nets_utils.py:485-498): forward / input gradient / parameter gradients
  conv_module_selu.npz           ConvolutionModule(64, 7, SELU): activation behind BatchNorm, forward + backward
  subsampling_odim40.npz         Conv2dSubsampling(20, 40): an output width that is not a multiple of 64
  subsampling6_odim48.npz        Conv2dSubsampling6(30, 48)
  e2e_rnn_vggblstm.npz           RNN E2E with etype vggblstm (stacked BLSTM without projections: BASELINE config 4's encoder)
  e2e_rnn_ss.npz                 RNN E2E (as e2e_rnn.npz) with sampling_probability 0.5 (rnn/decoders.py:249-254); the
                                 Python `random` stream is seeded with 7 right before the forward pass
  warmup_lr.npz                  espnet2 WarmupLR (schedulers/warmup_lr.py:10-53): the lr of 14 optimizer steps, warmup 5
  adadelta.npz                   torch.optim.Adadelta as asr.py:505-508 builds it (rho 0.95, eps 1e-8) with clip_grad_norm_(5)
                                 and one _adadelta_eps_decay(0.01) (asr_utils.py:517-528) after the third step
  e2e_rnn_batchbeam.npz          E2E.recognize_batch -> Decoder.recognize_beam_batch (e2e_asr.py:394-445, rnn/decoders.py:632-974)
                                 on the three utterances of e2e_rnn.npz (the model of seed 31): attention only, joint CTC (the CPU
                                 rule: CTC scores for the int(1.5 * beam) best labels of a hypothesis), CTC + RNNLM, nbest 2

Usage: python oracle/gen_golden_r3.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import install_stubs, save  # noqa: E402


def sd_np(m, pre):
    return {pre + k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def grads_np(m):
    return {"grad/" + k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    install_stubs()
    sys.path.insert(0, a.ref)
    out = lambda n: os.path.join(a.out, n)  # noqa: E731
    torch.set_num_threads(4)

    from espnet.nets.pytorch_backend.conformer.convolution import ConvolutionModule
    from espnet.nets.pytorch_backend.nets_utils import get_activation
    from espnet.nets.pytorch_backend.transformer.embedding import PositionalEncoding
    from espnet.nets.pytorch_backend.transformer.positionwise_feed_forward import PositionwiseFeedForward
    from espnet.nets.pytorch_backend.transformer.subsampling import Conv2dSubsampling, Conv2dSubsampling6

    # ---- a8 with the remaining activations ----
    torch.manual_seed(81)
    for name in ("hardtanh", "tanh", "selu"):
        ff = PositionwiseFeedForward(64, 96, 0.0, get_activation(name))
        x = (2.0 * torch.randn(2, 6, 64)).requires_grad_(True)      # wide enough to saturate hardtanh on many units
        y = ff(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        save(out("ffn_%s.npz" % name), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, **sd_np(ff, "sd/"), **grads_np(ff))

    # ---- a9 with an activation other than swish / relu behind the BatchNorm ----
    torch.manual_seed(91)
    cm_ = ConvolutionModule(64, 7, get_activation("selu"))
    cm_.norm.weight.data.uniform_(0.5, 1.5)
    cm_.norm.bias.data.uniform_(-0.3, 0.3)
    sd0 = sd_np(cm_, "sd/")
    x = torch.randn(3, 13, 64, requires_grad=True)
    cm_.train()
    y = cm_(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    save(out("conv_module_selu.npz"), x=x.detach(), y=y.detach(), gy=gy, gx=x.grad, **sd0, **grads_np(cm_))

    # ---- a3 at output widths that are not multiples of 64 ----
    torch.manual_seed(35)
    for fname, cls, idim, odim, T in (("subsampling_odim40.npz", Conv2dSubsampling, 20, 40, 37),
                                      ("subsampling6_odim48.npz", Conv2dSubsampling6, 30, 48, 41)):
        sub = cls(idim, odim, 0.0, PositionalEncoding(odim, 0.0))
        x = torch.randn(2, T, idim)
        m = torch.ones(2, 1, T, dtype=torch.bool)
        m[1, 0, T - 9:] = False
        y, ym = sub(x, m)
        gy = torch.randn_like(y)
        y.backward(gy)
        save(out(fname), x=x, mask=m, y=y.detach(), ymask=ym, gy=gy, **sd_np(sub, "sd/"), **grads_np(sub))

    # ---- a20 with scheduled sampling ----
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

    torch.manual_seed(31)
    m = RnnE2E(12, 7, rnn_args(sampling_probability=0.5))
    m.train()
    sd0 = sd_np(m, "sd/")
    g = torch.Generator().manual_seed(3)
    xs = torch.randn(3, 41, 12, generator=g)
    ilens = torch.tensor([41, 33, 20])
    ys = torch.randint(1, 6, (3, 6), generator=g)
    ys[1, 4:] = -1
    ys[2, 3:] = -1
    xs = xs * (torch.arange(41).view(1, -1, 1) < ilens.view(-1, 1, 1))
    random.seed(7)
    loss = m(xs, ilens, ys)
    loss.backward()
    random.seed(7)
    coins = [random.random() for _ in range(6)]      # the draws the decoder made (steps 1..6), for the record
    save(out("e2e_rnn_ss.npz"), xs=xs, ilens=ilens, ys=ys, loss=float(loss), loss_att=float(m.loss_att),
         loss_ctc=float(m.loss_ctc), acc=float(m.acc), coins=np.asarray(coins), **sd0, **grads_np(m))

    # ---- a20 with BASELINE config 4's encoder type: VGG + stacked (non-projected) BLSTM (rnn/encoders.py:103-162) ----
    torch.manual_seed(34)
    m = RnnE2E(12, 7, rnn_args(etype="vggblstm", elayers=2, eunits=12, eprojs=10))
    m.train()
    sd0 = sd_np(m, "sd/")
    hs, hlens, _ = m.enc(xs, ilens)
    loss = m(xs, ilens, ys)
    loss.backward()
    save(out("e2e_rnn_vggblstm.npz"), xs=xs, ilens=ilens, ys=ys, hs_pad=hs.detach(), hlens=np.asarray(hlens, dtype=np.int64),
         loss=float(loss), loss_att=float(m.loss_att), loss_ctc=float(m.loss_ctc), acc=float(m.acc), **sd0, **grads_np(m))

    # ---- a19 / a20: vectorised batch beam search of the RNN decoder ----
    from espnet.nets.pytorch_backend.lm.default import ClassifierWithState, RNNLM
    torch.manual_seed(31)
    m = RnnE2E(12, 7, rnn_args())
    m.eval()
    sd0 = sd_np(m, "sd/")
    torch.manual_seed(131)
    lm = ClassifierWithState(RNNLM(7, 1, 8, None, "lstm", 0.0)).eval()
    res = dict(sd_np(lm, "rlm/"))
    feats = [xs[b, : int(ilens[b])].numpy() for b in range(3)]
    with torch.no_grad():
        for tag, kw, use_lm in (("b3", dict(beam_size=3, ctc_weight=0.0, penalty=0.0), False),
                                ("b3ctc", dict(beam_size=3, ctc_weight=0.5, penalty=0.1), False),
                                ("b2lm", dict(beam_size=2, ctc_weight=0.3, penalty=0.0, lm_weight=0.4), True),
                                ("b4len", dict(beam_size=4, ctc_weight=0.3, penalty=0.2, maxlenratio=0.4, minlenratio=0.1), False)):
            ra = argparse.Namespace(**dict(dict(nbest=2, maxlenratio=0.0, minlenratio=0.0, lm_weight=0.0, ctc_window_margin=0), **kw))
            nb = m.recognize_batch(feats, ra, rnn_args().char_list, lm if use_lm else None)
            res["bb_%s_n" % tag] = np.asarray([len(u) for u in nb], dtype=np.int64)
            res["bb_%s_scores" % tag] = np.asarray([float(np.asarray(h["score"]).reshape(-1)[0]) for u in nb for h in u], dtype=np.float64)
            res["bb_%s_lens" % tag] = np.asarray([len(h["yseq"]) for u in nb for h in u], dtype=np.int64)
            res["bb_%s_yseq" % tag] = np.asarray(sum([[int(t) for t in h["yseq"]] for u in nb for h in u], []), dtype=np.int64)
            print(tag, [[(h["yseq"], round(float(np.asarray(h["score"]).reshape(-1)[0]), 4)) for h in u] for u in nb])
    save(out("e2e_rnn_batchbeam.npz"), xs=xs, ilens=ilens, **sd0, **res)

    # ---- a18: WarmupLR ----
    from espnet2.schedulers.warmup_lr import WarmupLR
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.Adam([w], lr=0.002)
    sch = WarmupLR(opt, warmup_steps=5)
    lrs = []
    for _ in range(14):
        lrs.append(opt.param_groups[0]["lr"])        # the lr optimizer.step() uses
        w.grad = torch.ones(3)
        opt.step()
        sch.step()
    save(out("warmup_lr.npz"), lrs=np.asarray(lrs, dtype=np.float64), base_lr=0.002, warmup=5)

    # ---- a18: Adadelta as the RNN recipes use it ----
    from espnet.asr.asr_utils import _adadelta_eps_decay
    g = torch.Generator().manual_seed(18)
    n = 1003
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adadelta([pr], rho=0.95, eps=1e-8, weight_decay=0.0)

    class _U:
        def get_optimizer(self, _name):
            return opt

    class _T:
        updater = _U()

    grs, traj = [], []
    for step in range(6):
        gr = torch.randn(n, generator=g) * (30.0 if step == 2 else 0.5)
        grs.append(gr.clone())
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        traj.append(pr.detach().clone())
        if step == 2:
            _adadelta_eps_decay(_T(), 0.01)
    save(out("adadelta.npz"), p0=p0, grads=torch.stack(grs), traj=torch.stack(traj), eps_after=opt.param_groups[0]["eps"])
    print("round-3 fixtures written to", a.out)


if __name__ == "__main__":
    main()
