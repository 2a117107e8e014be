"""Transducer decoding: greedy search and the default beam search.
reference: espnet/nets/beam_search_transducer.py:23-237 (Hypothesis, BeamSearchTransducer.__init__/__call__/
sort_nbest/greedy_search/default_beam_search).  The control flow (hypothesis lists, expansion order, the prediction
network cache keyed by the label prefix) is host Python exactly as in the reference; every arithmetic step
(embedding, LSTM / GRU step, joint network, log-softmax) runs on the espnet_amd kernels.  The tsd / alsd / nsc
variants and LM fusion are not on the HIP path yet and raise."""
from dataclasses import dataclass
from typing import Any, Dict, List, Union

import torch

from .. import ops


@dataclass
class Hypothesis:
    """reference: beam_search_transducer.py:23-33"""

    score: float
    yseq: List[int]
    dec_state: Any
    y: List[torch.Tensor] = None
    lm_state: Union[Dict[str, Any], List[Any]] = None
    lm_scores: torch.Tensor = None


def _log_softmax(z):
    return ops.log_softmax_rows(z.reshape(1, -1).contiguous()).view(-1)


class BeamSearchTransducer:
    def __init__(self, decoder, beam_size, lm=None, lm_weight=0.1, search_type="default", max_sym_exp=2, u_max=50,
                 nstep=1, prefix_alpha=1, score_norm=True):
        self.decoder = decoder
        self.beam_size = beam_size
        self.hidden_size = decoder.dunits
        self.vocab_size = decoder.odim
        self.blank = decoder.blank
        if self.beam_size <= 1:
            self.search_algorithm = self.greedy_search
        elif search_type == "default":
            self.search_algorithm = self.default_beam_search
        else:
            raise NotImplementedError("search_type %r: greedy and 'default' are on the HIP path" % search_type)
        # lm: a ClassifierWithState (espnet_amd.nets.lm, as espnet/asr/pytorch_backend/asr.py passes it) whose
        # predict(state, tokens) -> (state, log-probs (1, V)); fused with lm_weight in the default search
        self.lm, self.lm_weight = lm, lm_weight
        self.max_sym_exp, self.u_max, self.nstep, self.prefix_alpha = max_sym_exp, u_max, nstep, prefix_alpha
        self.score_norm = score_norm

    def __call__(self, h):
        """h: encoded speech features (T_max, D_enc) -> 1-best Hypothesis (greedy) or sorted n-best list"""
        with torch.no_grad():
            return self.search_algorithm(h)

    def sort_nbest(self, hyps):
        if self.score_norm:
            return sorted(hyps, key=lambda x: x.score / len(x.yseq), reverse=True)
        return sorted(hyps, key=lambda x: x.score, reverse=True)

    def greedy_search(self, h):
        """reference: beam_search_transducer.py:130-162"""
        init_tensor = h.unsqueeze(0)
        dec_state = self.decoder.init_state(init_tensor)
        hyp = Hypothesis(score=0.0, yseq=[self.blank], dec_state=dec_state)
        cache = {}
        y, state, _ = self.decoder.score(hyp, cache, init_tensor)
        # the encoder side of the joint network does not depend on the hypothesis: one GEMM for all frames
        enc_proj = self.decoder.joint_network.project_enc(h)
        for i in range(h.shape[0]):
            ytu = _log_softmax(self.decoder.joint_network.joint_step(enc_proj[i], y[0]))
            logp, pred = torch.max(ytu, dim=-1)
            pred = int(pred)
            if pred != self.blank:
                hyp.yseq.append(pred)
                hyp.score += float(logp)
                hyp.dec_state = state
                y, state, _ = self.decoder.score(hyp, cache, init_tensor)
        return hyp

    def default_beam_search(self, h):
        """reference: beam_search_transducer.py:164-237"""
        beam = min(self.beam_size, self.vocab_size)
        beam_k = min(beam, (self.vocab_size - 1))
        init_tensor = h.unsqueeze(0)
        dec_state = self.decoder.init_state(init_tensor)
        kept_hyps = [Hypothesis(score=0.0, yseq=[self.blank], dec_state=dec_state)]
        cache = {}
        enc_proj = self.decoder.joint_network.project_enc(h)
        for t in range(h.shape[0]):
            hyps = kept_hyps
            kept_hyps = []
            while True:
                max_hyp = max(hyps, key=lambda x: x.score)
                hyps.remove(max_hyp)
                y, state, lm_tokens = self.decoder.score(max_hyp, cache, init_tensor)
                ytu = _log_softmax(self.decoder.joint_network.joint_step(enc_proj[t], y[0]))
                top_v, top_i = ytu[1:].topk(beam_k, dim=-1)
                cand = list(zip(top_v.tolist(), (top_i + 1).tolist())) + [(float(ytu[0]), self.blank)]
                if self.lm:
                    lm_state, lm_scores = self.lm.predict(max_hyp.lm_state, lm_tokens)
                    lm_host = lm_scores[0].tolist()
                for logp, k in cand:
                    new_hyp = Hypothesis(score=(max_hyp.score + float(logp)), yseq=max_hyp.yseq[:],
                                         dec_state=max_hyp.dec_state, lm_state=max_hyp.lm_state)
                    if k == self.blank:
                        kept_hyps.append(new_hyp)
                    else:
                        new_hyp.dec_state = state
                        new_hyp.yseq.append(int(k))
                        if self.lm:
                            new_hyp.lm_state = lm_state
                            new_hyp.score += self.lm_weight * lm_host[k]
                        hyps.append(new_hyp)
                hyps_max = float(max(hyps, key=lambda x: x.score).score)
                kept_most_prob = sorted([hyp for hyp in kept_hyps if hyp.score > hyps_max], key=lambda x: x.score)
                if len(kept_most_prob) >= beam:
                    kept_hyps = kept_most_prob
                    break
        return self.sort_nbest(kept_hyps)
