"""Attention decoder of the RNN path.  reference: espnet/nets/pytorch_backend/rnn/decoders.py:28-311
(Decoder.__init__/zero_state/rnn_forward/forward, lstm, single encoder, no scheduled sampling)."""
import math

import random

import numpy as np
import torch

from ... import functional as F_
from ... import ops
from ... import rnn_functional as R_
from ..modules import pad_list, th_accuracy


class LSTMCell(torch.nn.Module):
    """parameter container with torch.nn.LSTMCell's names; forward = x-projection GEMM + fused cell"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        k = 1.0 / math.sqrt(hidden_size)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = torch.nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = torch.nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))
        self.bias_hh = torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))

    def forward(self, x, state):
        h, c = state
        gx = F_.LinearFn.apply(x, self.weight_ih, self.bias_ih)
        return R_.LSTMCellFn.apply(gx, h, c, self.weight_hh, self.bias_hh)


class GRUCell(torch.nn.Module):
    """parameter container with torch.nn.GRUCell's names; forward(x, h) -> h'"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        k = 1.0 / math.sqrt(hidden_size)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = torch.nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = torch.nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = torch.nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh = torch.nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))

    def forward(self, x, h):
        gx = F_.LinearFn.apply(x, self.weight_ih, self.bias_ih)
        return R_.GRUCellFn.apply(gx, h, self.weight_hh, self.bias_hh)


class Decoder(torch.nn.Module):
    """reference: rnn/decoders.py:28-311"""

    def __init__(self, eprojs, odim, dtype, dlayers, dunits, sos, eos, att, verbose=0, char_list=None,
                 labeldist=None, lsm_weight=0.0, sampling_probability=0.0, dropout=0.0, context_residual=False,
                 replace_sos=False, num_encs=1):
        super().__init__()
        if dtype not in ("lstm", "gru"):
            raise NotImplementedError("dtype %r: lstm and gru decoders have HIP kernels" % dtype)
        if num_encs != 1 or replace_sos or labeldist is not None:
            raise NotImplementedError("multi-encoder / replace_sos / label-dist smoothing are outside the hot-path scope")
        self.dtype, self.dunits, self.dlayers, self.context_residual = dtype, dunits, dlayers, context_residual
        self.embed = torch.nn.Embedding(odim, dunits)
        cell = LSTMCell if dtype == "lstm" else GRUCell
        self.decoder = torch.nn.ModuleList([cell(dunits + eprojs, dunits)] + [cell(dunits, dunits) for _ in range(1, dlayers)])
        self.ignore_id = -1
        self.output = torch.nn.Linear(dunits + eprojs if context_residual else dunits, odim)
        self.loss = None
        self.att = att
        self.sos, self.eos, self.odim = sos, eos, odim
        self.verbose, self.char_list = verbose, char_list
        self.lsm_weight = lsm_weight
        self.sampling_probability = sampling_probability
        self.dropout = dropout
        self.num_encs = num_encs
        self.logzero = -10000000000.0
        self.salt_emb = ops.new_salt()
        self.salts = [ops.new_salt() for _ in range(dlayers)]

    def zero_state(self, hs_pad):
        return hs_pad.new_zeros(hs_pad.size(0), self.dunits)

    def _drop(self, k, x, step):
        # one salt per (layer, step): torch draws a fresh mask at every call of dropout_dec[k]
        return F_.dropout(x, self.dropout, self.salts[k] + 131 * (step + 1), self.training)

    def rnn_forward(self, ey, z_list, c_list, z_prev, c_prev, step=0):
        if self.dtype == "lstm":
            z_list[0], c_list[0] = self.decoder[0](ey, (z_prev[0], c_prev[0]))
            for i in range(1, self.dlayers):
                z_list[i], c_list[i] = self.decoder[i](self._drop(i - 1, z_list[i - 1], step), (z_prev[i], c_prev[i]))
        else:
            z_list[0] = self.decoder[0](ey, z_prev[0])
            for i in range(1, self.dlayers):
                z_list[i] = self.decoder[i](self._drop(i - 1, z_list[i - 1], step), z_prev[i])
        return z_list, c_list

    def forward(self, hs_pad, hlens, ys_pad, strm_idx=0, lang_ids=None):
        """hs_pad (B,T,D), hlens list[int], ys_pad (B,Lmax) padded with -1 -> (loss, acc, ppl)"""
        hlens = [int(v) for v in hlens]
        ys = [y[y != self.ignore_id] for y in ys_pad.cpu()]       # host-side label parsing (decoders.py:167);
        # pass ys_pad as a CPU tensor to keep this free of a device->host sync (needed under hipGraph capture)
        eos = ys[0].new([self.eos])
        sos = ys[0].new([self.sos])
        ys_in = [torch.cat([sos, y], dim=0) for y in ys]
        ys_out = [torch.cat([y, eos], dim=0) for y in ys]
        dev = hs_pad.device
        ys_in_pad = ops.h2d_cached("ys_in", pad_list(ys_in, self.eos).numpy(), dev)
        ys_out_pad = ops.h2d_cached("ys_out", pad_list(ys_out, self.ignore_id).numpy(), dev)
        batch, olength = ys_out_pad.size(0), ys_out_pad.size(1)
        c_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        z_list = [self.zero_state(hs_pad) for _ in range(self.dlayers)]
        z_all = []
        att_w = None
        att = self.att[min(strm_idx, len(self.att) - 1)]
        att.reset()
        eys = F_.dropout(R_.PlainEmbedFn.apply(ys_in_pad, self.embed.weight, -1), self.dropout, self.salt_emb,
                         self.training)
        for i in range(olength):
            att_c, att_w = att(hs_pad, hlens, self._drop(0, z_list[0], i + olength), att_w)
            if i > 0 and random.random() < self.sampling_probability:
                # scheduled sampling (decoders.py:249-254): the previous step's own argmax token is embedded instead of the
                # reference label - one host-side coin per step as in the reference; the argmax and the embedding lookup
                # stay on the device (no host read-back)
                z_out = F_.LinearFn.apply(z_all[-1], self.output.weight, self.output.bias)
                tok = ops.argmax_rows(z_out.detach().contiguous()).view(batch, 1).long()
                z_emb = F_.dropout(R_.PlainEmbedFn.apply(tok, self.embed.weight, -1), self.dropout, self.salt_emb + 977 * (i + 1),
                                   self.training)
                ey = torch.cat((z_emb[:, 0, :], att_c), dim=1)
            else:
                ey = torch.cat((eys[:, i, :], att_c), dim=1)
            z_list, c_list = self.rnn_forward(ey, z_list, c_list, z_list, c_list, step=i)
            top = self._drop(self.dlayers - 1, z_list[-1], i + 2 * olength)
            z_all.append(torch.cat((top, att_c), dim=-1) if self.context_residual else top)
        z_all = torch.stack(z_all, dim=1).view(batch * olength, -1)
        y_all = F_.LinearFn.apply(z_all, self.output.weight, self.output.bias)
        n_tok = sum(len(y) for y in ys_out)
        ce, correct = F_.LabelSmoothingLossFn.apply(y_all.view(batch, olength, -1), ys_out_pad, 0.0, self.ignore_id,
                                                    float(n_tok))
        ppl = torch.exp(ce.detach())
        # -1: eos, which is removed in the loss computation (decoders.py:271-272)
        self.loss = F_.ScaleFn.apply(ce, float(np.mean([len(x) for x in ys_in]) - 1))
        acc = th_accuracy(correct, ys_out_pad, self.ignore_id)
        return self.loss, acc, ppl


def _recognize_beam(self, h, lpz, recog_args, char_list=None, rnnlm=None, strm_idx=0):
    """reference: rnn/decoders.py:313-605 (single encoder).  h (T, eprojs) encoder states, lpz (T, odim) CTC
    log-posteriors or None -> n-best list of {"score", "yseq"} dicts.  Hypotheses are expanded one by one as in the
    reference (attention state, decoder cells and the pre-beam are per hypothesis); the CTC prefix scores of a
    hypothesis' pre-beam come from one eamd_ctc_prefix_score launch instead of the numpy CTCPrefixScore."""
    from argparse import Namespace

    from ..beam_search import end_detect
    from ..ctc_prefix_score import CTCPrefixScorer
    CTC_SCORING_RATIO = 1.5
    dev = h.device
    att = self.att[min(strm_idx, len(self.att) - 1)]
    hx = h.unsqueeze(0)
    c_list = [self.zero_state(hx) for _ in range(self.dlayers)]
    z_list = [self.zero_state(hx) for _ in range(self.dlayers)]
    att.reset()
    beam, penalty = recog_args.beam_size, recog_args.penalty
    ctc_weight = getattr(recog_args, "ctc_weight", False)
    lm_weight = getattr(recog_args, "lm_weight", 0.0)
    y = self.sos
    maxlen = h.size(0)
    if recog_args.maxlenratio != 0:
        maxlen = max(1, int(recog_args.maxlenratio * maxlen))
    minlen = int(recog_args.minlenratio * maxlen)
    hyp = {"score": 0.0, "yseq": [y], "c_prev": c_list, "z_prev": z_list, "a_prev": None}
    if rnnlm:
        hyp["rnnlm_prev"] = None
    scorer = None
    if lpz is not None:
        scorer = CTCPrefixScorer(None, self.eos)
        scorer.logp = lpz.detach().contiguous()
        r0 = torch.full((lpz.size(0), 2), self.logzero, device=dev)
        r0[:, 1] = torch.cumsum(scorer.logp[:, 0], 0)
        hyp["ctc_state_prev"] = r0
        hyp["ctc_score_prev"] = 0.0
        ctc_beam = min(lpz.shape[-1], int(beam * CTC_SCORING_RATIO)) if ctc_weight != 1.0 else lpz.shape[-1]
    hyps, ended_hyps = [hyp], []
    with torch.no_grad():
        for i in range(maxlen):
            hyps_best_kept = []
            for hyp in hyps:
                vy = torch.full((1,), hyp["yseq"][i], dtype=torch.long, device=dev)
                ey = R_.PlainEmbedFn.apply(vy, self.embed.weight, -1)
                att_c, att_w = att(hx, [h.size(0)], hyp["z_prev"][0], hyp["a_prev"])
                ey = torch.cat((ey, att_c), dim=1)
                z_list, c_list = self.rnn_forward(ey, [None] * self.dlayers, [None] * self.dlayers, hyp["z_prev"],
                                                  hyp["c_prev"])
                top = torch.cat((z_list[-1], att_c), dim=-1) if self.context_residual else z_list[-1]
                logits = F_.LinearFn.apply(top, self.output.weight, self.output.bias)
                local_att_scores = ops.log_softmax_rows(logits.contiguous())
                if rnnlm:
                    rnnlm_state, local_lm_scores = rnnlm.predict(hyp["rnnlm_prev"], vy)
                    local_scores = local_att_scores + lm_weight * local_lm_scores
                else:
                    local_scores = local_att_scores
                if scorer is not None:
                    _, local_best_ids = torch.topk(local_att_scores, ctc_beam, dim=1)
                    ids = local_best_ids.to(torch.int32)
                    last = torch.tensor([hyp["yseq"][-1]], dtype=torch.int32, device=dev)
                    olen = torch.tensor([len(hyp["yseq"]) - 1], dtype=torch.int32, device=dev)
                    psi, r_new = ops.ctc_prefix_score(scorer.logp, hyp["ctc_state_prev"].unsqueeze(0), ids, last, olen,
                                                      0, self.eos)
                    local_scores = (1.0 - ctc_weight) * local_att_scores[:, local_best_ids[0]] \
                        + ctc_weight * (psi - hyp["ctc_score_prev"])
                    if rnnlm:
                        local_scores = local_scores + lm_weight * local_lm_scores[:, local_best_ids[0]]
                    local_best_scores, joint_best_ids = torch.topk(local_scores, beam, dim=1)
                    local_best_ids = local_best_ids[:, joint_best_ids[0]]
                    joint = joint_best_ids[0].tolist()
                    psi_host = psi[0].tolist()
                else:
                    local_best_scores, local_best_ids = torch.topk(local_scores, beam, dim=1)
                best_scores, best_ids = local_best_scores[0].tolist(), local_best_ids[0].tolist()
                for j in range(beam):
                    new_hyp = {"z_prev": z_list[:], "c_prev": c_list[:], "a_prev": att_w,
                               "score": hyp["score"] + best_scores[j], "yseq": hyp["yseq"] + [int(best_ids[j])]}
                    if rnnlm:
                        new_hyp["rnnlm_prev"] = rnnlm_state
                    if scorer is not None:
                        new_hyp["ctc_state_prev"] = r_new[0, joint[j]]
                        new_hyp["ctc_score_prev"] = psi_host[joint[j]]
                    hyps_best_kept.append(new_hyp)
                hyps_best_kept = sorted(hyps_best_kept, key=lambda x: x["score"], reverse=True)[:beam]
            hyps = hyps_best_kept
            if i == maxlen - 1:      # force <eos> so that something ends (decoders.py:561-564)
                for hyp in hyps:
                    hyp["yseq"].append(self.eos)
            remained_hyps = []
            for hyp in hyps:
                if hyp["yseq"][-1] == self.eos:
                    if len(hyp["yseq"]) > minlen:       # shorter ones are dropped (decoders.py:569-580)
                        hyp["score"] += (i + 1) * penalty
                        if rnnlm:
                            hyp["score"] += lm_weight * rnnlm.final(hyp["rnnlm_prev"])
                        ended_hyps.append(hyp)
                else:
                    remained_hyps.append(hyp)
            if end_detect(ended_hyps, i) and recog_args.maxlenratio == 0.0:
                break
            hyps = remained_hyps
            if len(hyps) == 0:
                break
    nbest_hyps = sorted(ended_hyps, key=lambda x: x["score"], reverse=True)[: min(len(ended_hyps), recog_args.nbest)]
    if len(nbest_hyps) == 0:     # decoders.py:607-619: retry with a smaller minimum length
        recog_args = Namespace(**vars(recog_args))
        recog_args.minlenratio = max(0.0, recog_args.minlenratio - 0.1)
        return self.recognize_beam(h, lpz, recog_args, char_list, rnnlm)
    return nbest_hyps


Decoder.recognize_beam = _recognize_beam


def _recognize_beam_batch(self, h, hlens, lpz, recog_args, char_list=None, rnnlm=None, normalize_score=True, strm_idx=0,
                          ctc_scoring_num=None):
    """Vectorised beam search over a BATCH of utterances (reference: rnn/decoders.py:632-974, single encoder, no CTC window):
    h (B, Tmax, eprojs), hlens (B,), lpz (B, Tmax, odim) CTC log-posteriors or None -> per utterance an n-best list of
    {"yseq", "score", "vscore"}.

    All B x beam hypotheses advance together: ONE embedding / attention / decoder-cell / output step per position over the
    (B * beam) rows, a top-`beam` per utterance over its beam x odim candidates (<eos> is scored aside: a hypothesis ends
    when its <eos> score beats the worst survivor), CTC prefix scores in the reference's full-matrix form (CTCPrefixScoreTH:
    log-zero outside the scored labels, <eos> from the last frame, blank excluded; one eamd_ctc_prefix_score launch per
    utterance and step on the utterance's own frames - padding frames change nothing in that recursion).  Prefixes, scores and
    every state stay on the device; the host reads one small tensor per step for the ended-hypothesis bookkeeping and the
    per-utterance end detection (e2e_asr_common.end_detect), as the reference does.
    ctc_scoring_num: labels per hypothesis that receive a CTC score; None = the reference's rule for device tensors (0: the whole
    vocabulary); its CPU rule is int(1.5 * beam) (what the fixtures were recorded with)."""
    from ..beam_search import BeamSearch, end_detect
    from ..ctc_prefix_score import CTCPrefixScorer
    dev = h.device
    hlens = [int(v) for v in hlens]
    B, beam, V = len(hlens), int(recog_args.beam_size), self.odim
    penalty = float(recog_args.penalty)
    ctc_weight = float(getattr(recog_args, "ctc_weight", 0.0)) if lpz is not None else 0.0
    att_weight = 1.0 - ctc_weight
    lm_weight = float(getattr(recog_args, "lm_weight", 0.0))
    if int(getattr(recog_args, "ctc_window_margin", 0)) > 0:
        raise NotImplementedError("ctc_window_margin > 0 (attention-windowed CTC scoring) is outside the hot-path scope")
    att = self.att[min(strm_idx, len(self.att) - 1)]
    n = B * beam
    tmask = torch.arange(h.size(1), device=dev)[None, :] < torch.tensor(hlens, device=dev)[:, None]
    h = h * tmask.unsqueeze(-1)                                                       # mask_by_length(h, hlens, 0.0)
    max_hlen = max(hlens)
    maxlen = max_hlen if recog_args.maxlenratio == 0 else max(1, int(recog_args.maxlenratio * max_hlen))
    minlen = int(recog_args.minlenratio * max_hlen)
    exp_h = h.unsqueeze(1).expand(B, beam, h.size(1), h.size(2)).reshape(n, h.size(1), h.size(2)).contiguous()
    exp_hlens = [hlens[b] for b in range(B) for _ in range(beam)]
    z_prev = [exp_h.new_zeros(n, self.dunits) for _ in range(self.dlayers)]
    c_prev = [exp_h.new_zeros(n, self.dunits) for _ in range(self.dlayers)]
    a_prev = None
    att.reset()
    yseq = torch.full((n, maxlen + 2), self.eos, dtype=torch.long, device=dev)
    yseq[:, 0] = self.sos
    vscores = torch.zeros(B, beam, device=dev)
    base = (torch.arange(B, device=dev) * beam).view(B, 1)
    scorers, c_s, c_r = None, None, None
    if lpz is not None:
        if ctc_scoring_num is None:
            ctc_scoring_num = 0
        snum = min(int(ctc_scoring_num) if att_weight > 0.0 else 0, V)
        scorers = []
        for b in range(B):
            sc = CTCPrefixScorer(None, self.eos)
            sc.logp = lpz[b, : hlens[b]].detach().contiguous()
            scorers.append(sc)
        c_s = [torch.zeros(beam, device=dev) for _ in range(B)]
        c_r = []
        for b in range(B):
            r0 = torch.full((hlens[b], 2), self.logzero, device=dev)
            r0[:, 1] = torch.cumsum(scorers[b].logp[:, 0], 0)
            c_r.append(r0.unsqueeze(0).expand(beam, hlens[b], 2).contiguous())
    lm_state = None
    stop = [False] * B
    ended = [[] for _ in range(B)]
    with torch.no_grad():
        for i in range(maxlen):
            L = i + 1
            vy = yseq[:, i].contiguous()
            ey = R_.PlainEmbedFn.apply(vy, self.embed.weight, -1)
            att_c, att_w = att(exp_h, exp_hlens, z_prev[0], a_prev)
            ey = torch.cat((ey, att_c), dim=1)
            z_list, c_list = self.rnn_forward(ey, [None] * self.dlayers, [None] * self.dlayers, z_prev, c_prev)
            top = torch.cat((z_list[-1], att_c), dim=-1) if self.context_residual else z_list[-1]
            logits = F_.LinearFn.apply(top, self.output.weight, self.output.bias)
            local = att_weight * ops.log_softmax_rows(logits.contiguous())
            if rnnlm is not None:
                lm_state, lm_scores = rnnlm.buff_predict(lm_state, vy, n)
                local = local + lm_weight * lm_scores
            c_full, c_rnew, c_idmap = None, None, None
            if scorers is not None:
                local[:, 0] = self.logzero                                      # blank is never chosen
                part_ids = torch.topk(local, snum, dim=-1)[1] if snum > 0 else None
                c_full, c_rnew, c_idmap = [], [], []
                ys = yseq[:, :L]
                for b in range(B):
                    rows = slice(b * beam, (b + 1) * beam)
                    ids = part_ids[rows] if part_ids is not None else torch.arange(V, device=dev).unsqueeze(0).expand(beam, V)
                    last = ys[rows, -1].to(torch.int32).contiguous()
                    olen = torch.full((beam,), L - 1, dtype=torch.int32, device=dev)
                    psi, r_new = ops.ctc_prefix_score(scorers[b].logp, c_r[b], ids.to(torch.int32).contiguous(), last, olen, 0,
                                                      self.eos)
                    full = torch.full((beam, V), self.logzero, device=dev)
                    full.scatter_(1, ids.long(), psi)
                    full[:, self.eos] = torch.logsumexp(c_r[b][:, -1, :], dim=-1)
                    full[:, 0] = self.logzero
                    idmap = torch.full((beam, V), -1, dtype=torch.long, device=dev)
                    idmap.scatter_(1, ids.long(), torch.arange(ids.shape[1], device=dev).expand(beam, -1))
                    local[rows] += ctc_weight * (full - c_s[b][:, None])
                    c_full.append(full)
                    c_rnew.append(r_new)
                    c_idmap.append(idmap)
            local = local.view(B, beam, V)
            if i == 0:
                local[:, 1:, :] = self.logzero                                  # all slots hold the same <sos> prefix
            eos_vscores = local[:, :, self.eos] + vscores
            cand = vscores.unsqueeze(-1).expand(B, beam, V).clone()
            cand[:, :, self.eos] = self.logzero
            cand = (cand + local).view(B, beam * V)
            best_scores, best_ids = torch.topk(cand, beam, dim=1)
            tok = best_ids % V
            src = (best_ids // V + base).view(-1)                                # row of the (B * beam) batch each survivor extends
            y_prev = yseq
            yseq = yseq.index_select(0, src)
            yseq[:, L] = tok.view(-1)
            vscores = best_scores
            a_prev = att_w.index_select(0, src) if torch.is_tensor(att_w) else None
            if not torch.is_tensor(att_w):
                raise NotImplementedError("recognize_beam_batch: attention types whose state is not one weight tensor")
            z_prev = [z.index_select(0, src) for z in z_list]
            c_prev = [c.index_select(0, src) for c in c_list] if self.dtype == "lstm" else c_prev
            # ---- ended hypotheses and end detection: one device -> host copy per step ----
            if i >= minlen:
                host = torch.cat([eos_vscores, best_scores], dim=1).cpu()         # [B, 2 * beam]
                pen = (i + 1) * penalty
                yp_host = yn_host = None
                for b in range(B):
                    if stop[b]:
                        continue
                    thr = float(host[b, 2 * beam - 1])
                    for j in range(beam):
                        k = b * beam + j
                        val, seq = None, None
                        if float(host[b, j]) > thr:
                            if L <= hlens[b]:
                                if yp_host is None:
                                    yp_host = y_prev[:, :L].cpu()
                                val, seq = float(host[b, j]) + pen, yp_host[k].tolist()
                        elif i == maxlen - 1:
                            if yn_host is None:
                                yn_host = yseq[:, : L + 1].cpu()
                            val, seq = float(host[b, beam + j]) + pen, yn_host[k].tolist()
                        if val:                                                  # the reference's truthiness test on the score
                            seq = seq + [self.eos]
                            if rnnlm is not None:
                                val += lm_weight * float(rnnlm.final(lm_state, index=k))
                            ended[b].append({"yseq": seq, "vscore": val, "score": val})
            stop = [stop[b] or end_detect(ended[b], i) for b in range(B)]
            if all(stop):
                break
            if rnnlm is not None:
                lm_state = BeamSearch._tree_index(lm_state, src)
            if scorers is not None:
                for b in range(B):
                    hb = best_ids[b] // V
                    tb = tok[b]
                    j = c_idmap[b][hb, tb].clamp_min(0)
                    c_s[b] = c_full[b][hb, tb]
                    c_r[b] = c_rnew[b][hb, j]
    out = []
    for b in range(B):
        hyps = ended[b] if ended[b] else [{"yseq": [self.sos, self.eos], "score": -float("inf"), "vscore": -float("inf")}]
        if normalize_score:
            for x in hyps:
                x["score"] = x["score"] / len(x["yseq"])
        out.append(sorted(hyps, key=lambda x: x["score"], reverse=True)[: min(len(hyps), int(recog_args.nbest))])
    return out


Decoder.recognize_beam_batch = _recognize_beam_batch


def decoder_for(args, odim, sos, eos, att, labeldist):
    """reference: rnn/decoders.py:1199-1218"""
    return Decoder(args.eprojs, odim, args.dtype, args.dlayers, args.dunits, sos, eos, att, args.verbose,
                   args.char_list, labeldist, args.lsm_weight, args.sampling_probability, args.dropout_rate_decoder,
                   getattr(args, "context_residual", False), getattr(args, "replace_sos", False),
                   getattr(args, "num_encs", 1))
