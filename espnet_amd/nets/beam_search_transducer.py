"""Transducer decoding: greedy search, the default beam search, time-synchronous (tsd) and alignment-length
synchronous (alsd) decoding and the N-step constrained search (nsc), with optional RNNLM shallow fusion.
reference: espnet/nets/beam_search_transducer.py:23-462 (Hypothesis, BeamSearchTransducer.__init__/__call__/
sort_nbest/greedy_search/default_beam_search/time_sync_decoding/align_length_sync_decoding/nsc_beam_search, :23-662).  The control flow (hypothesis lists, expansion order, the prediction
network cache keyed by the label prefix) is host Python exactly as in the reference; every arithmetic step
(embedding, LSTM / GRU step, joint network, log-softmax) runs on the espnet_amd kernels - batched over the beam in the
tsd / alsd / nsc searches."""
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
        elif search_type == "tsd":
            self.search_algorithm = self.time_sync_decoding
        elif search_type == "alsd":
            self.search_algorithm = self.align_length_sync_decoding
        elif search_type == "nsc":
            self.search_algorithm = self.nsc_beam_search
        else:
            raise NotImplementedError("search_type %r: greedy, default, tsd, alsd, nsc" % search_type)
        # lm: a ClassifierWithState (espnet_amd.nets.lm, as espnet/asr/pytorch_backend/asr.py passes it) whose
        # predict(state, tokens) -> (state, log-probs (1, V)); fused with lm_weight in the default search
        self.lm, self.lm_weight = lm, lm_weight
        self.max_sym_exp, self.u_max, self.nstep, self.prefix_alpha = max_sym_exp, u_max, nstep, prefix_alpha
        self.score_norm = score_norm

    def __call__(self, h):
        """h: encoded speech features (T_max, D_enc) -> 1-best Hypothesis (greedy) or sorted n-best list"""
        if hasattr(self.decoder, "att"):          # rnnt-att: forget the previous utterance's encoder projections (:105-109)
            self.decoder.att[0].reset()
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

    # ---- LM state plumbing of the batched searches (transducer/utils.py:96-178; RNNLM states {c, h} per layer) --------
    def _lm_init(self):
        lm_model = self.lm.predictor
        p = next(lm_model.parameters())
        state = {"h": [torch.zeros(lm_model.n_units, device=p.device, dtype=p.dtype) for _ in range(len(lm_model.rnn))]}
        if lm_model.typ == "lstm":
            state["c"] = [torch.zeros(lm_model.n_units, device=p.device, dtype=p.dtype) for _ in range(len(lm_model.rnn))]
        return state

    @staticmethod
    def _lm_batch(states):
        return {k: [torch.stack([s[k][layer] for s in states]) for layer in range(len(states[0][k]))] for k in states[0]}

    @staticmethod
    def _lm_select(states, idx):
        return {k: [v[layer][idx] for layer in range(len(v))] for k, v in states.items()}

    def _joint_logp(self, h_enc, beam_y):
        """h_enc (n, D_enc) or (1, D_enc), beam_y (n, D_dec) -> log-softmax of the joint outputs (n, V): the n
        (frame, hypothesis) pairs as one batch through lin_enc / lin_dec / the joint kernel / lin_out"""
        from .. import functional as F_
        from .. import rnn_functional as R_
        jn = self.decoder.joint_network
        n = beam_y.shape[0]
        enc = jn.project_enc(h_enc)
        if enc.shape[0] != n:
            enc = enc.expand(n, -1).contiguous()
        d = F_.LinearFn.apply(beam_y.contiguous(), jn.lin_dec.weight, None)
        z = R_.JointFn.apply(enc.view(n, 1, -1).contiguous(), d.view(n, 1, -1), jn.act_id)
        logits = F_.LinearFn.apply(z.reshape(n, -1), jn.lin_out.weight, jn.lin_out.bias)
        return ops.log_softmax_rows(logits.contiguous())

    def time_sync_decoding(self, h):
        """reference: beam_search_transducer.py:238-350 (https://ieeexplore.ieee.org/document/9053040)"""
        import numpy as np
        beam = min(self.beam_size, self.vocab_size)
        init_tensor = h.unsqueeze(0)
        beam_state = self.decoder.init_state(torch.zeros((beam, self.hidden_size), device=h.device))
        B = [Hypothesis(yseq=[self.blank], score=0.0, dec_state=self.decoder.select_state(beam_state, 0))]
        if self.lm:
            B[0].lm_state = self._lm_init()
        cache = {}
        for t in range(h.shape[0]):
            A = []
            C = B
            h_enc = h[t].unsqueeze(0)
            for v in range(self.max_sym_exp):
                D = []
                beam_y, beam_state, beam_lm_tokens = self.decoder.batch_score(C, beam_state, cache, init_tensor)
                beam_logp = self._joint_logp(h_enc, beam_y)
                top_v, top_i = beam_logp[:, 1:].topk(beam, dim=-1)
                blank_lp, top_v, top_i = beam_logp[:, 0].tolist(), top_v.tolist(), (top_i + 1).tolist()
                seq_A = [hy.yseq for hy in A]
                for i, hyp in enumerate(C):
                    if hyp.yseq not in seq_A:
                        A.append(Hypothesis(score=(hyp.score + blank_lp[i]), yseq=hyp.yseq[:], dec_state=hyp.dec_state,
                                            lm_state=hyp.lm_state))
                    else:
                        pos = seq_A.index(hyp.yseq)
                        A[pos].score = np.logaddexp(A[pos].score, (hyp.score + blank_lp[i]))
                if v < self.max_sym_exp:
                    if self.lm:
                        beam_lm_states, beam_lm_scores = self.lm.buff_predict(
                            self._lm_batch([c.lm_state for c in C]), beam_lm_tokens, len(C))
                        lm_host = beam_lm_scores.tolist()
                    for i, hyp in enumerate(C):
                        for logp, k in zip(top_v[i], top_i[i]):
                            new_hyp = Hypothesis(score=(hyp.score + float(logp)), yseq=(hyp.yseq + [int(k)]),
                                                 dec_state=self.decoder.select_state(beam_state, i), lm_state=hyp.lm_state)
                            if self.lm:
                                new_hyp.score += self.lm_weight * lm_host[i][k]
                                new_hyp.lm_state = self._lm_select(beam_lm_states, i)
                            D.append(new_hyp)
                C = sorted(D, key=lambda x: x.score, reverse=True)[:beam]
            B = sorted(A, key=lambda x: x.score, reverse=True)[:beam]
        return self.sort_nbest(B)

    def align_length_sync_decoding(self, h):
        """reference: beam_search_transducer.py:352-462 (https://ieeexplore.ieee.org/document/9053040)"""
        import numpy as np
        beam = min(self.beam_size, self.vocab_size)
        h_length = int(h.size(0))
        u_max = min(self.u_max, (h_length - 1))
        init_tensor = h.unsqueeze(0)
        beam_state = self.decoder.init_state(torch.zeros((beam, self.hidden_size), device=h.device))
        B = [Hypothesis(yseq=[self.blank], score=0.0, dec_state=self.decoder.select_state(beam_state, 0))]
        final = []
        if self.lm:
            B[0].lm_state = self._lm_init()
        cache = {}
        for i in range(h_length + u_max):
            A, B_, h_states = [], [], []
            for hyp in B:
                u = len(hyp.yseq) - 1
                t = i - u + 1
                if t > (h_length - 1):
                    continue
                B_.append(hyp)
                h_states.append((t, h[t]))
            if B_:
                beam_y, beam_state, beam_lm_tokens = self.decoder.batch_score(B_, beam_state, cache, init_tensor)
                h_enc = torch.stack([hs[1] for hs in h_states])
                beam_logp = self._joint_logp(h_enc, beam_y)
                top_v, top_i = beam_logp[:, 1:].topk(beam, dim=-1)
                blank_lp, top_v, top_i = beam_logp[:, 0].tolist(), top_v.tolist(), (top_i + 1).tolist()
                if self.lm:
                    beam_lm_states, beam_lm_scores = self.lm.buff_predict(
                        self._lm_batch([b.lm_state for b in B_]), beam_lm_tokens, len(B_))
                    lm_host = beam_lm_scores.tolist()
                for j, hyp in enumerate(B_):
                    new_hyp = Hypothesis(score=(hyp.score + blank_lp[j]), yseq=hyp.yseq[:], dec_state=hyp.dec_state,
                                         lm_state=hyp.lm_state)
                    A.append(new_hyp)
                    if h_states[j][0] == (h_length - 1):
                        final.append(new_hyp)
                    for logp, k in zip(top_v[j], top_i[j]):
                        new_hyp = Hypothesis(score=(hyp.score + float(logp)), yseq=(hyp.yseq[:] + [int(k)]),
                                             dec_state=self.decoder.select_state(beam_state, j), lm_state=hyp.lm_state)
                        if self.lm:
                            new_hyp.score += self.lm_weight * lm_host[j][k]
                            new_hyp.lm_state = self._lm_select(beam_lm_states, j)
                        A.append(new_hyp)
                B = sorted(A, key=lambda x: x.score, reverse=True)[:beam]
                # recombine_hyps (transducer/utils.py:181-203): scores of equal sequences are merged into the first one,
                # the list itself is returned unchanged
                firsts = []
                for hyp in B:
                    seqs = [f.yseq for f in firsts if f.yseq]
                    if hyp.yseq in seqs:
                        f = firsts[seqs.index(hyp.yseq)]
                        f.score = np.logaddexp(f.score, hyp.score)
                    else:
                        firsts.append(hyp)
        if final:
            return self.sort_nbest(final)
        return B

    def nsc_beam_search(self, h):
        """N-step constrained beam search.  reference: beam_search_transducer.py:464-662
        (https://arxiv.org/pdf/2002.03577.pdf as modified there)"""
        import numpy as np
        beam = min(self.beam_size, self.vocab_size)
        beam_k = min(beam, (self.vocab_size - 1))
        jn = self.decoder.joint_network
        init_tensor = h.unsqueeze(0)
        beam_state = self.decoder.init_state(torch.zeros((beam, self.hidden_size), device=h.device))
        init_tokens = [Hypothesis(yseq=[self.blank], score=0.0, dec_state=self.decoder.select_state(beam_state, 0))]
        cache = {}
        beam_y, beam_state, beam_lm_tokens = self.decoder.batch_score(init_tokens, beam_state, cache, init_tensor)
        state = self.decoder.select_state(beam_state, 0)
        lm_state = lm_scores = None
        if self.lm:
            beam_lm_states, beam_lm_scores = self.lm.buff_predict(None, beam_lm_tokens, 1)
            lm_state = self._lm_select(beam_lm_states, 0)
            lm_scores = beam_lm_scores[0]
        kept_hyps = [Hypothesis(yseq=[self.blank], score=0.0, dec_state=state, y=[beam_y[0]], lm_state=lm_state,
                                lm_scores=lm_scores)]
        enc_proj = jn.project_enc(h)

        def is_prefix(x, pref):
            return len(pref) < len(x) and all(pref[i] == x[i] for i in range(len(pref)))

        for t in range(h.shape[0]):
            hyps = sorted(kept_hyps, key=lambda x: len(x.yseq), reverse=True)
            kept_hyps = []
            h_enc = h[t].unsqueeze(0)
            for j in range(len(hyps) - 1):       # prefix search: a longer hypothesis also collects its prefixes' mass
                for i in range((j + 1), len(hyps)):
                    if is_prefix(hyps[j].yseq, hyps[i].yseq) and \
                            (len(hyps[j].yseq) - len(hyps[i].yseq)) <= self.prefix_alpha:
                        next_id = len(hyps[i].yseq)
                        ytu = _log_softmax(jn.joint_step(enc_proj[t], hyps[i].y[-1]))
                        curr_score = hyps[i].score + float(ytu[hyps[j].yseq[next_id]])
                        for k in range(next_id, (len(hyps[j].yseq) - 1)):
                            ytu = _log_softmax(jn.joint_step(enc_proj[t], hyps[j].y[k]))
                            curr_score += float(ytu[hyps[j].yseq[k + 1]])
                        hyps[j].score = np.logaddexp(hyps[j].score, curr_score)
            S, V = [], []
            for n in range(self.nstep):
                beam_y = torch.stack([hyp.y[-1] for hyp in hyps])
                beam_logp = self._joint_logp(h_enc, beam_y)
                top_v, top_i = beam_logp[:, 1:].topk(beam_k, dim=-1)
                blank_lp, top_v, top_i = beam_logp[:, 0].tolist(), top_v.tolist(), (top_i + 1).tolist()
                if self.lm:
                    lm_host = torch.stack([hyp.lm_scores for hyp in hyps]).tolist()
                for i, hyp in enumerate(hyps):
                    for logp, k in list(zip(top_v[i], top_i[i])) + [(blank_lp[i], self.blank)]:
                        new_hyp = Hypothesis(yseq=hyp.yseq[:], score=(hyp.score + float(logp)), y=hyp.y[:],
                                             dec_state=hyp.dec_state, lm_state=hyp.lm_state, lm_scores=hyp.lm_scores)
                        if k == self.blank:
                            S.append(new_hyp)
                        else:
                            new_hyp.yseq.append(int(k))
                            if self.lm:
                                new_hyp.score += self.lm_weight * float(lm_host[i][k])
                        V.append(new_hyp)       # blank extensions too: `substract` drops them again (same yseq as a parent)
                V = sorted(V, key=lambda x: x.score, reverse=True)
                V = [v for v in V if not any(v.yseq == hy.yseq for hy in hyps)][:beam]
                beam_state = self.decoder.create_batch_states(beam_state, [v.dec_state for v in V], [v.yseq for v in V])
                beam_y, beam_state, beam_lm_tokens = self.decoder.batch_score(V, beam_state, cache, init_tensor)
                if self.lm:
                    beam_lm_states, beam_lm_scores = self.lm.buff_predict(self._lm_batch([v.lm_state for v in V]),
                                                                           beam_lm_tokens, len(V))
                if n < (self.nstep - 1):
                    for i, v in enumerate(V):
                        v.y.append(beam_y[i])
                        v.dec_state = self.decoder.select_state(beam_state, i)
                        if self.lm:
                            v.lm_state = self._lm_select(beam_lm_states, i)
                            v.lm_scores = beam_lm_scores[i]
                    hyps = V[:]
                else:
                    last_blank = self._joint_logp(h_enc, beam_y)[:, 0].tolist()
                    for i, v in enumerate(V):
                        if self.nstep != 1:
                            v.score += float(last_blank[i])
                        v.y.append(beam_y[i])
                        v.dec_state = self.decoder.select_state(beam_state, i)
                        if self.lm:
                            v.lm_state = self._lm_select(beam_lm_states, i)
                            v.lm_scores = beam_lm_scores[i]
            kept_hyps = sorted((S + V), key=lambda x: x.score, reverse=True)[:beam]
        return self.sort_nbest(kept_hyps)
