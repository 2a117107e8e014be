"""Transducer decoding on the device primitives: greedy search and the default (Graves) beam search.

What the searches compute is pinned by the reference (espnet/nets/beam_search_transducer.py:130-237: one symbol per
frame at most in the greedy search; the A / B hypothesis sets of Graves 2012 in the default search, RNNLM shallow
fusion added to the non-blank extensions) and by the recorded hypotheses in tests/golden/transducer_*.npz.  How it is
computed is this package's own design:

  * label prefixes live in a TRIE (`_Prefixes`): a hypothesis is (score, node).  Everything that depends on the
    label sequence only - the prediction network's output and state, the LM state and scores - is stored ONCE per
    node under its integer id, so two hypotheses that reach the same sequence share it;
  * the joint network runs on ROWS: `JointNetwork.joint_rows` takes n (frame, node) pairs and returns n log-softmax
    rows from one lin_dec GEMM, one fused add + activation, one lin_out GEMM and one log-softmax launch;
  * greedy search: the prediction output only changes when a symbol is emitted, so a whole CHUNK of frames is scored
    against the current node in one batch, the first non-blank frame is located on the device and only that index
    and token come back to the host - launches scale with emitted symbols, not with frames;
  * default beam search: at the start of a frame all carried hypotheses are scored in one batch (prediction steps for
    nodes first seen + joint rows + top-k, one device-to-host copy); children created during the frame are scored
    lazily, again all pending ones in one batch, when the first of them is selected for expansion.

Prediction networks take part through one of two protocols: `step(tokens, states)` (batched; DecoderRNNT) or the
reference's per-hypothesis plug-in method `score(hyp, cache, init_tensor)` (any TransducerDecoderInterface).
The time-synchronous / alignment-length-synchronous / N-step-constrained searches of the reference are outside the
hot-path scope (greedy + default) and are not provided."""
from dataclasses import dataclass, field
from typing import Any, List

import torch

from .. import ops


@dataclass
class Hypothesis:
    """result record (the fields the reference's recognisers read: score, yseq; the state fields stay empty)"""

    score: float
    yseq: List[int]
    dec_state: Any = None
    lm_state: Any = None
    y: Any = None
    lm_scores: Any = None


class _Prefixes:
    """trie of label sequences; node 0 = [blank]"""

    def __init__(self, blank):
        self.parent, self.token, self.kids = [-1], [blank], [{}]

    def child(self, node, tok):
        c = self.kids[node].get(tok)
        if c is None:
            c = len(self.parent)
            self.parent.append(node)
            self.token.append(tok)
            self.kids.append({})
            self.kids[node][tok] = c
        return c

    def labels(self, node):
        out = []
        while node >= 0:
            out.append(self.token[node])
            node = self.parent[node]
        return out[::-1]


@dataclass
class _View:
    """what a TransducerDecoderInterface.score() reads of a hypothesis"""

    yseq: List[int]
    dec_state: Any
    score: float = 0.0
    lm_state: Any = None


@dataclass
class _Pred:
    """per-node results of the prediction network (and of the LM, filled on first expansion)"""

    out: dict = field(default_factory=dict)       # node -> (dunits,) output
    state: dict = field(default_factory=dict)     # node -> state AFTER consuming the node's token
    lm_state: dict = field(default_factory=dict)
    lm_logp: dict = field(default_factory=dict)   # node -> host list of LM log-probabilities of the next token


class BeamSearchTransducer:
    def __init__(self, decoder, beam_size, lm=None, lm_weight=0.1, search_type="default", max_sym_exp=2, u_max=50,
                 nstep=1, prefix_alpha=1, score_norm=True, frame_chunk=32):
        self.decoder = decoder
        self.beam_size = beam_size
        self.vocab_size = decoder.odim
        self.blank = decoder.blank
        if self.blank != 0:
            raise NotImplementedError("blank id must be 0 (the non-blank top-k is taken over ids 1..V-1)")
        if beam_size > 1 and search_type != "default":
            raise NotImplementedError("search_type %r: the greedy (beam_size <= 1) and default searches are provided" % search_type)
        self.lm, self.lm_weight = lm, lm_weight
        self.score_norm = score_norm
        self.frame_chunk = frame_chunk
        self.batched = hasattr(decoder, "step")

    @ops.inference_call
    def __call__(self, h):
        """h: encoder states of one utterance (T, D_enc) -> 1-best Hypothesis (greedy) or the sorted n-best list"""
        if hasattr(self.decoder, "att"):          # rnnt-att: forget the previous utterance's encoder projections
            self.decoder.att[0].reset()
        with torch.no_grad():
            return self._greedy(h) if self.beam_size <= 1 else self._default(h)

    # ---- prediction network, once per trie node ---------------------------------------------------------------------
    def _start(self, h):
        self._tree = _Prefixes(self.blank)
        self._pred = _Pred()
        self._init_tensor = h.unsqueeze(0)
        self._root_state = self.decoder.init_state(self._init_tensor)
        self._ext_cache = {}

    def _ensure_pred(self, nodes):
        """prediction-network output / state for every node of `nodes` that has none yet (parents always have)"""
        todo = [n for n in dict.fromkeys(nodes) if n not in self._pred.out]
        if not todo:
            return
        tr, pr, dec = self._tree, self._pred, self.decoder
        if self.batched:
            dev = self._init_tensor.device
            toks = torch.tensor([tr.token[n] for n in todo], dtype=torch.long, device=dev)
            prev = [pr.state[tr.parent[n]] if tr.parent[n] >= 0 else dec.unbatch_state(self._root_state, 0) for n in todo]
            y, new = dec.step(toks, dec.batch_states(prev))
            for i, n in enumerate(todo):
                pr.out[n] = y[i]
                pr.state[n] = dec.unbatch_state(new, i)
        else:
            for n in todo:
                prev = pr.state[tr.parent[n]] if tr.parent[n] >= 0 else self._root_state
                y, st, _ = dec.score(_View(tr.labels(n), prev), self._ext_cache, self._init_tensor)
                pr.out[n] = y.reshape(-1)
                pr.state[n] = st

    def _lm_step(self, node):
        """LM log-probabilities of the token after `node` (RNNLM shallow fusion), once per node"""
        pr = self._pred
        if node not in pr.lm_logp:
            par = self._tree.parent[node]
            prev = pr.lm_state[par] if par >= 0 else None
            tok = torch.tensor([self._tree.token[node]], dtype=torch.long, device=self._init_tensor.device)
            st, logp = self.lm.predict(prev, tok)
            pr.lm_state[node] = st
            pr.lm_logp[node] = logp[0].tolist()
        return pr.lm_logp[node]

    def _rows(self, enc_rows, nodes):
        """log-softmax rows of the joint network for n (encoder row, node) pairs -> (n, V)"""
        y = torch.stack([self._pred.out[n] for n in nodes])
        return ops.log_softmax_rows(self.decoder.joint_network.joint_rows(enc_rows, y))

    # ---- greedy -----------------------------------------------------------------------------------------------------
    def _greedy(self, h):
        self._start(h)
        jn = self.decoder.joint_network
        enc = jn.project_enc(h)
        T = enc.shape[0]
        node, score, t = 0, 0.0, 0
        self._ensure_pred([0])
        while t < T:
            n = min(self.frame_chunk, T - t)
            logp = ops.log_softmax_rows(jn.joint_rows(enc[t:t + n], self._pred.out[node].unsqueeze(0)))
            best = ops.argmax_rows(logp).long()                                     # (n,)
            emit = (best != self.blank)
            # first emitting frame of the chunk, its token and log-probability: one small copy to the host
            first = torch.where(emit, torch.arange(n, device=best.device), torch.full((), n, device=best.device)).min()
            idx = first.clamp(max=n - 1)
            rec = torch.stack([first.float(), best[idx].float(), logp[idx, best[idx]]]).tolist()
            f = int(rec[0])
            if f >= n:                     # only blanks in this chunk
                t += n
                continue
            node = self._tree.child(node, int(rec[1]))
            score += rec[2]
            self._ensure_pred([node])
            t += f + 1                     # at most one symbol per frame (reference greedy_search)
        return Hypothesis(score=score, yseq=self._tree.labels(node))

    # ---- default beam search ----------------------------------------------------------------------------------------
    def _default(self, h):
        self._start(h)
        enc = self.decoder.joint_network.project_enc(h)
        beam = min(self.beam_size, self.vocab_size)
        beam_k = min(beam, self.vocab_size - 1)
        carried = [(0.0, 0)]                    # set B of the previous frame: (score, node)
        for t in range(enc.shape[0]):
            frontier = carried                  # set A
            carried = []
            scored = {}                         # node -> (blank logp, [(logp, token)] best non-blank extensions) at frame t

            def score_pending(first):
                """joint rows for `first` and every other node of the frontier that has none at this frame"""
                nodes = [first] + [n for _s, n in frontier if n not in scored and n != first]
                nodes = list(dict.fromkeys(nodes))
                self._ensure_pred(nodes)
                logp = self._rows(enc[t].unsqueeze(0).expand(len(nodes), -1), nodes)
                top_v, top_i = logp[:, 1:].topk(beam_k, dim=-1)
                host = torch.cat([logp[:, :1], top_v, (top_i + 1).float()], dim=1).tolist()
                for n, row in zip(nodes, host):
                    scored[n] = (row[0], list(zip(row[1:1 + beam_k], (int(v) for v in row[1 + beam_k:]))))

            while True:
                j = max(range(len(frontier)), key=lambda i: frontier[i][0])          # first maximum, as max() over a list
                s_best, n_best = frontier.pop(j)
                if n_best not in scored:
                    score_pending(n_best)
                blank_lp, ext = scored[n_best]
                lm_lp = self._lm_step(n_best) if self.lm else None
                for lp, tok in ext:
                    s = s_best + lp
                    if lm_lp is not None:
                        s += self.lm_weight * lm_lp[tok]
                    frontier.append((s, self._tree.child(n_best, tok)))
                carried.append((s_best + blank_lp, n_best))
                bound = max(s for s, _n in frontier)
                ahead = sorted((c for c in carried if c[0] > bound), key=lambda c: c[0])
                if len(ahead) >= beam:
                    carried = ahead
                    break
        hyps = [Hypothesis(score=s, yseq=self._tree.labels(n)) for s, n in carried]
        if self.score_norm:
            return sorted(hyps, key=lambda x: x.score / len(x.yseq), reverse=True)
        return sorted(hyps, key=lambda x: x.score, reverse=True)
