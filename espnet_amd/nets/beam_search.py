"""Joint CTC/attention(/LM) beam search for one utterance, every beam step batched through the HIP kernels.

reference: espnet/nets/beam_search.py:36-458 (BeamSearch: constructor, scorer dictionaries, pre-beam on the
full score, partial scorers on the pre-beam only, top-`beam` over all expansions, <eos> handling and
final_score in post_process, end detection), espnet/nets/e2e_asr_common.py:21-51 (end_detect), legacy
E2E.recognize options (e2e_asr_transformer.py:259-477: ctc_weight, lm_weight, penalty, maxlenratio,
minlenratio, nbest).

Same constructor and `forward(x, maxlenratio, minlenratio) -> List[Hypothesis]` as the reference.  Where the
reference's BeamSearch scores hypothesis by hypothesis (`score` / `score_partial`), one step here is: one
`batch_score` per full scorer over all running hypotheses, one top-k for the pre-beam, one CTC prefix-score
launch, one flat top-k, and one device->host copy of the selected (hypothesis, token, scores) rows.
Utterances are independent: decode many by running one search per utterance on separate streams / GPUs
(SURVEY.md §8e "replicas only").
"""
import math
import os
from itertools import chain
from typing import Any, Dict, NamedTuple

import torch

from ..ops import inference_call as _inference_call
from .ctc_prefix_score import CTCPrefixScorer, LengthBonus
from .scorer_interface import PartialScorerInterface, ScorerInterface


class Hypothesis(NamedTuple):
    """reference: beam_search.py:20-33 (yseq is an int64 tensor that starts with <sos>)"""

    yseq: torch.Tensor
    score: Any = 0.0
    scores: Dict[str, Any] = dict()
    states: Dict[str, Any] = dict()

    def asdict(self):
        return self._replace(yseq=self.yseq.tolist(), score=float(self.score),
                             scores={k: float(v) for k, v in self.scores.items()})._asdict()


def end_detect(ended_hyps, i, M=3, D_end=math.log(1 * math.exp(-10))):
    """reference: e2e_asr_common.py:21-51"""
    if len(ended_hyps) == 0:
        return False
    best = max(h["score"] for h in ended_hyps)
    count = 0
    for m in range(M):
        same = [h["score"] for h in ended_hyps if len(h["yseq"]) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


class _NoDynStep(Exception):
    """a beam step that cannot read its step index from the device (BeamSearch._batch_step(dyn=...)): one graph per step stays"""


class _OutsideCandidates(Exception):
    """raised by the host side of a "full"-mode search whose candidate selection kernel picked a log-zero continuation"""


def _end_detect_sl(ended, i, M=3, D_end=math.log(1 * math.exp(-10))):
    """end_detect on (score, length) pairs (the same test without building a dict per hypothesis per step)"""
    if not ended:
        return False
    best = max(s for s, _ in ended)
    count = 0
    for m in range(M):
        same = [s for s, n in ended if n == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


class BeamSearch(torch.nn.Module):
    # how partial scorers report: "ids" = scores of the pre-beam ids only, everything else is dropped
    # (beam_search.py:226-262); "full" = (n, V) matrices (batch_beam_search.py:221-231)
    partial_mode = "ids"
    apply_final_score = True

    def __init__(self, scorers, weights, beam_size, vocab_size, sos, eos, token_list=None, pre_beam_ratio=1.5,
                 pre_beam_score_key=None):
        super().__init__()
        self.weights = weights
        self.scorers, self.full_scorers, self.part_scorers = dict(), dict(), dict()
        self.nn_dict = torch.nn.ModuleDict()          # so that .to() / .eval() reach the scorer modules
        for k, v in scorers.items():
            if weights.get(k, 0) == 0 or v is None:
                continue
            assert isinstance(v, ScorerInterface), f"{k} ({type(v)}) does not implement ScorerInterface"
            self.scorers[k] = v
            (self.part_scorers if isinstance(v, PartialScorerInterface) else self.full_scorers)[k] = v
            if isinstance(v, torch.nn.Module):
                self.nn_dict[k] = v
        self.sos, self.eos, self.token_list = sos, eos, token_list
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        self.beam_size, self.n_vocab = beam_size, vocab_size
        if (pre_beam_score_key is not None and pre_beam_score_key != "full"
                and pre_beam_score_key not in self.full_scorers):
            raise KeyError(f"{pre_beam_score_key} is not found in {self.full_scorers}")
        self.pre_beam_score_key = pre_beam_score_key
        self.do_pre_beam = (pre_beam_score_key is not None and self.pre_beam_size < vocab_size
                            and len(self.part_scorers) > 0)

    # ---- hypothesis bookkeeping (host side, as in the reference) -------------------------------------
    def init_hyp(self, x):
        states = {k: d.batch_init_state(x) if hasattr(d, "batch_init_state") else d.init_state(x)
                  for k, d in self.scorers.items()}
        return [Hypothesis(yseq=torch.tensor([self.sos], dtype=torch.int64), score=0.0,
                           scores={k: 0.0 for k in self.scorers}, states=states)]

    @staticmethod
    def append_token(xs, x):
        return torch.cat((xs, torch.tensor([int(x)], dtype=xs.dtype)))

    # ---- one step ------------------------------------------------------------------------------------
    def search(self, running_hyps, x):
        """running hypotheses (all of one length) -> the `beam_size` best one-token extensions, best first"""
        n, V, dev = len(running_hyps), self.n_vocab, x.device
        ys = torch.stack([h.yseq for h in running_hyps]).to(dev)
        xs = x.unsqueeze(0).expand(n, *x.shape)
        weighted = torch.zeros(n, V, device=dev, dtype=torch.float32)
        scores, states = {}, {}
        for k, d in self.full_scorers.items():
            if hasattr(d, "batch_score"):
                scores[k], states[k] = d.batch_score(ys, [h.states[k] for h in running_hyps], xs)
            else:      # a plain ScorerInterface (e.g. the attention RNN decoder): hypothesis by hypothesis
                sc, st = zip(*[d.score(ys[i], running_hyps[i].states[k], x) for i in range(n)])
                scores[k], states[k] = torch.stack(sc), list(st)
            weighted += self.weights[k] * scores[k]
        part_ids = None
        if self.do_pre_beam:
            pre = weighted if self.pre_beam_score_key == "full" else scores[self.pre_beam_score_key]
            part_ids = torch.topk(pre, self.pre_beam_size, dim=-1)[1]
        part_scores, part_states = {}, {}
        if self.part_scorers:
            if self.partial_mode == "full":
                for k, d in self.part_scorers.items():
                    part_scores[k], part_states[k] = d.batch_score_partial(
                        ys, part_ids, [h.states[k] for h in running_hyps], x)
                    weighted += self.weights[k] * part_scores[k]
            else:
                ids = part_ids if part_ids is not None else torch.arange(V, device=dev).unsqueeze(0).expand(n, V)
                local = torch.zeros(n, ids.shape[1], device=dev, dtype=torch.float32)
                for k, d in self.part_scorers.items():
                    part_scores[k], part_states[k] = d.score_partial_batch(
                        ys, ids, [h.states[k] for h in running_hyps], x)
                    local += self.weights[k] * part_scores[k]
                if part_ids is not None:   # tokens outside the pre-beam are dropped (beam_search.py:252-262)
                    kept = torch.full_like(weighted, -float("inf"))
                    kept.scatter_(1, ids, torch.gather(weighted, 1, ids) + local)
                    weighted = kept
                else:
                    weighted += local
        weighted += torch.tensor([float(h.score) for h in running_hyps], dtype=torch.float32).to(dev)[:, None]

        # global top-`beam` over all (hypothesis, token) expansions; everything the host needs in one copy
        k_sel = min(self.beam_size, n * V)
        top_s, top_i = weighted.view(-1).topk(k_sel)
        hyp_i, tok_i = top_i // V, top_i % V
        cols = [top_s, hyp_i.float(), tok_i.float()]
        names = list(scores.keys())
        for k in names:
            cols.append(scores[k][hyp_i, tok_i])
        pos = None
        for k in part_scores:
            if self.partial_mode == "full":
                cols.append(part_scores[k][hyp_i, tok_i])
            else:
                if pos is None:
                    pos = (ids[hyp_i] == tok_i[:, None]).float().argmax(-1) if part_ids is not None else tok_i
                cols.append(part_scores[k][hyp_i, pos])
        if pos is not None:
            cols.append(pos.float())
        host = torch.stack(cols, dim=1).cpu()
        out = []
        for row in host.tolist():
            if not math.isfinite(row[0]):
                continue
            a, tok = int(row[1]), int(row[2])
            prev = running_hyps[a]
            new_scores = dict(prev.scores)
            c = 3
            for k in names:
                new_scores[k] = float(prev.scores[k]) + row[c]
                c += 1
            for k in part_scores:
                new_scores[k] = float(prev.scores[k]) + row[c]
                c += 1
            new_states = {k: self.full_scorers[k].select_state(states[k], a) for k in names}
            for k in part_scores:
                if self.partial_mode == "full":
                    new_states[k] = self.part_scorers[k].select_state(part_states[k], a, tok)
                else:
                    new_states[k] = self.part_scorers[k].select_state(part_states[k], (a, int(row[-1])))
            out.append(Hypothesis(yseq=self.append_token(prev.yseq, tok), score=row[0], scores=new_scores,
                                  states=new_states))
        return out

    def post_process(self, i, maxlen, maxlenratio, running_hyps, ended_hyps):
        """reference: beam_search.py:407-458"""
        if i == maxlen - 1:      # force <eos> at the last position so that something ends
            running_hyps = [h._replace(yseq=self.append_token(h.yseq, self.eos)) for h in running_hyps]
        remained = []
        for hyp in running_hyps:
            if int(hyp.yseq[-1]) == self.eos:
                if self.apply_final_score:     # e.g. a word LM adds its final <eos> score
                    for k, d in chain(self.full_scorers.items(), self.part_scorers.items()):
                        s = d.final_score(hyp.states[k])
                        hyp.scores[k] += s
                        hyp = hyp._replace(score=hyp.score + self.weights[k] * s)
                ended_hyps.append(hyp)
            else:
                remained.append(hyp)
        return remained

    # ---- the same search with the hypotheses on the device ----------------------------------------------
    # `beam_size` slots hold prefixes [n, maxlen + 2], accumulated scores, per-scorer scores and every scorer's state as a
    # BATCHED tree (score_tree / CTC (s [n], r [n, T, 2])); a step selects with index_select / gather on device indices, dead
    # slots (ended or never filled) carry -inf and can never be selected again, and what the host needs of a step - scores,
    # tokens, the prefixes of the slots that ended - is logged in device tensors and fetched ONCE every `sync_every` steps
    # (one device -> host copy), where the reference's end detection is replayed step by step and the search is cut at the
    # step it would have stopped at.  Steps run beyond that point are discarded; results are those of the host loop.
    device_loop = True
    sync_every = 8
    candidate_select = True       # the selection of a step on the pre-beam candidates (eamd_beam_select); tests flip it
    step_kernel = True            # selection + bookkeeping of a pre-beam step in one launch (eamd_beam_step); tests flip it
    ctc_psi_parallel = True       # candidates scored by eamd_ctc_prefix_psi, survivors' states by eamd_ctc_prefix_state; tests flip it
    ctc_side_stream = "capture"   # ... the latter on a second stream beside the next step's decoder stack: True, False, or only where
                                  # it pays: more than 2048 frames (the frame-by-frame recursion, 130 us at 249 frames; up to 2048 the
                                  # states are a parallel scan of a few us) in a captured step graph (the fork and the join cost
                                  # ~25 us of queue switches per replay; an eager step is bound by the host's launches)

    def _ctc_stream(self, dev):
        st = getattr(self, "_ctc_side", None)
        if st is None or st.device != dev:
            st = self._ctc_side = torch.cuda.Stream(device=dev)
        return st

    def _device_loop_ok(self, x):
        return (self.device_loop and x.is_cuda and all(hasattr(d, "score_tree") for d in self.full_scorers.values())
                and all(isinstance(d, CTCPrefixScorer) for d in self.part_scorers.values()) and len(self.part_scorers) <= 1
                and len(self.full_scorers) <= 4)          # eamd_beam_finish carries up to four full scorers

    def _reorder(self, d, tree, idx):
        """a full scorer's batched state behind a selection: the scorer's own re-ordering where it has one (the Transformer decoder
        keeps key / value caches that are never moved, only a slot table is), else index_select on every tensor of the tree"""
        return d.reorder_tree(tree, idx) if hasattr(d, "reorder_tree") else self._tree_index(tree, idx)

    @staticmethod
    def _tree_index(tree, idx):
        if tree is None:
            return None
        if torch.is_tensor(tree):
            return tree.index_select(0, idx)
        if isinstance(tree, dict):
            return {k: BeamSearch._tree_index(v, idx) for k, v in tree.items()}
        if isinstance(tree, (list, tuple)):
            return type(tree)(BeamSearch._tree_index(v, idx) for v in tree)
        raise TypeError(type(tree))

    def _forward_device(self, x, maxlenratio, minlenratio):
        from .. import ops
        T, V, n, dev = x.shape[0], self.n_vocab, self.beam_size, x.device
        maxlen = T if maxlenratio == 0 else max(1, int(maxlenratio * T))
        NEG = -float("inf")
        names = list(self.full_scorers.keys())
        pname = next(iter(self.part_scorers), None)
        ctc = self.part_scorers[pname] if pname is not None else None
        allk = names + ([pname] if pname is not None else [])
        yseq = torch.full((n, maxlen + 2), self.eos, dtype=torch.int64, device=dev)
        yseq[:, 0] = self.sos
        hyp = torch.full((n,), NEG, device=dev, dtype=torch.float32)
        hyp[0] = 0.0
        sc = {k: torch.zeros(n, device=dev, dtype=torch.float32) for k in allk}
        trees = {k: None for k in names}
        for k, d in self.full_scorers.items():       # scorers that prepare per-utterance tensors (none of the tree scorers keeps one)
            if hasattr(d, "batch_init_state"):
                d.batch_init_state(x)
        if ctc is not None:
            s0, r0 = ctc.init_state(x)
            c_s = torch.zeros(n, device=dev, dtype=torch.float32)
            c_r = r0.unsqueeze(0).expand(n, *r0.shape).contiguous()
        xs = x.unsqueeze(0).expand(n, *x.shape)
        x1 = x.unsqueeze(0)            # for scorers with shared_memory_ok: ONE memory for the n hypotheses
        ended, pending, stop_at = [], [], None
        arange_v = torch.arange(V, device=dev).unsqueeze(0).expand(n, V) if (ctc is not None and not self.do_pre_beam) else None

        def flush():
            """fetch the logged steps, replay the reference's bookkeeping; returns True when the search is over"""
            nonlocal pending
            if not pending:
                return False
            host = torch.stack([p for p in pending]).cpu()          # [steps, n, 3 + len(allk) + maxlen + 2]
            pending = []
            for row in host:
                i = int(row[0, 0])
                alive = 0
                for slot in row.tolist():
                    top_s, tok = slot[1], int(slot[2])
                    if not math.isfinite(top_s):
                        continue
                    L = i + 2
                    seq = [int(v) for v in slot[3 + len(allk): 3 + len(allk) + L]]
                    if i == maxlen - 1:
                        seq.append(self.eos)
                    if seq[-1] == self.eos:
                        scores = {k: slot[3 + j] for j, k in enumerate(allk)}
                        if self.apply_final_score:
                            for k, d in chain(self.full_scorers.items(), self.part_scorers.items()):
                                f = float(d.final_score(None)) if not hasattr(d, "final_tree") else float(d.final_tree(None))
                                scores[k] += f
                                top_s += self.weights[k] * f
                        ended.append(Hypothesis(yseq=torch.tensor(seq, dtype=torch.int64), score=top_s, scores=scores, states={}))
                    else:
                        alive += 1
                if maxlenratio == 0.0 and end_detect([h.asdict() for h in ended], i):
                    return True
                if alive == 0:
                    return True
            return False

        with torch.no_grad():
            for i in range(maxlen):
                L = i + 1
                ys = yseq[:, :L]
                weighted = torch.zeros(n, V, device=dev, dtype=torch.float32)
                logps, newtrees = {}, {}
                for k, d in self.full_scorers.items():
                    logps[k], newtrees[k] = d.score_tree(ys, trees[k], x1 if getattr(d, "shared_memory_ok", False) else xs)
                    weighted += self.weights[k] * logps[k]
                part_ids = None
                if self.do_pre_beam:
                    pre = weighted if self.pre_beam_score_key == "full" else logps[self.pre_beam_score_key]
                    part_ids = ops.topk_rows(pre.contiguous(), self.pre_beam_size)[1]
                if ctc is not None:
                    last = ys[:, -1].to(torch.int32).contiguous()
                    olen = torch.full((n,), L - 1, dtype=torch.int32, device=dev)
                    if self.partial_mode == "full":
                        ids = part_ids if part_ids is not None else torch.arange(V, device=dev).unsqueeze(0).expand(n, V)
                        psi, r_new = ops.ctc_prefix_score(ctc.logp, c_r, ids.to(torch.int32).contiguous(), last, olen, ctc.blank, ctc.eos)
                        full = torch.full((n, V), -10000000000.0, device=dev, dtype=torch.float32)
                        full.scatter_(1, ids.long(), psi)
                        full[:, ctc.eos] = torch.logsumexp(c_r[:, -1, :], dim=-1)
                        full[:, ctc.blank] = -10000000000.0
                        idmap = torch.full((n, V), -1, dtype=torch.int64, device=dev)
                        idmap.scatter_(1, ids.long(), torch.arange(ids.shape[1], device=dev).expand(n, -1))
                        c_local = full - c_s[:, None]
                        weighted += self.weights[pname] * c_local
                    else:
                        ids = part_ids if part_ids is not None else arange_v
                        psi, r_new = ops.ctc_prefix_score(ctc.logp, c_r, ids.to(torch.int32).contiguous(), last, olen, ctc.blank, ctc.eos)
                        c_local = psi - c_s[:, None]
                        if part_ids is not None:
                            kept = torch.full_like(weighted, NEG)
                            kept.scatter_(1, ids, torch.gather(weighted, 1, ids) + self.weights[pname] * c_local)
                            weighted = kept
                        else:
                            weighted += self.weights[pname] * c_local
                weighted += hyp[:, None]
                s1, i1 = ops.topk_rows(weighted, n)                      # per slot, then among the n x n (see _batch_step)
                top_s, i2 = (v.view(-1) for v in ops.topk_rows(s1.view(1, n * n), n))
                top_i = (i2 // n) * V + i1.view(-1)[i2]
                hyp_i, tok_i = top_i // V, top_i % V
                for k in names:
                    sc[k] = sc[k][hyp_i] + logps[k][hyp_i, tok_i]
                    trees[k] = self._reorder(self.full_scorers[k], newtrees[k], hyp_i)
                if ctc is not None:
                    if self.partial_mode == "full":
                        sc[pname] = sc[pname][hyp_i] + c_local[hyp_i, tok_i]
                        j = idmap[hyp_i, tok_i].clamp_min(0)
                        c_s, c_r = full[hyp_i, tok_i], r_new[hyp_i, j]
                    else:
                        pos = (ids[hyp_i] == tok_i[:, None]).float().argmax(-1) if part_ids is not None else tok_i
                        sc[pname] = sc[pname][hyp_i] + c_local[hyp_i, pos]
                        c_s, c_r = psi[hyp_i, pos], r_new[hyp_i, pos]
                yseq = yseq.index_select(0, hyp_i)
                yseq[:, L] = tok_i
                finite = torch.isfinite(top_s)
                done = finite & (tok_i == self.eos) if i < maxlen - 1 else finite
                rec = torch.cat([torch.full((n, 1), float(i), device=dev), top_s[:, None], tok_i[:, None].float()]
                                + [sc[k][:, None] for k in allk] + [yseq.float()], dim=1)
                pending.append(rec)
                hyp = torch.where(done | ~finite, torch.full_like(top_s, NEG), top_s)
                if len(pending) >= self.sync_every or i == maxlen - 1:
                    if flush():
                        break
        nbest = sorted(ended, key=lambda h: float(h.score), reverse=True)
        if len(nbest) == 0:
            return [] if minlenratio < 0.1 else self.forward(x, maxlenratio, max(0.0, minlenratio - 0.1))
        return nbest

    @_inference_call
    def forward_batch(self, xs, maxlenratio=0.0, minlenratio=0.0):
        """Several utterances in ONE search: xs = list of (T_b, D) encoder outputs -> list of n-best lists (what forward() returns for
        each utterance alone).  B x beam slots share every launch of a beam step - one batch_score per scorer over all slots with the
        padded frames of shorter utterances masked in the source attention, one CTC prefix-score launch for all utterances
        (eamd_ctc_prefix_score_batch), one top-`beam` per utterance - and the host reads the step log once per `sync_every` steps,
        replaying ended-hypothesis bookkeeping and end detection per utterance.  An utterance that has finished keeps its slots (dead)
        until the last one finishes.  Falls back to one forward() per utterance when the scorers cannot keep batched states.
        With `graph_steps` the steps of a search are hipGraph replays (see _StepGraphs below)."""
        B = len(xs)
        if B == 0:
            return []
        from .. import ops
        ops.zero_arena_off()                    # a search never takes slices of a training step's zero arena
        ok = self._device_loop_ok(xs[0]) and minlenratio == 0.0
        if not ok:
            return [self.forward(x, maxlenratio, minlenratio) for x in xs]
        try:
            return self._forward_batch(xs, maxlenratio)
        except _OutsideCandidates:
            # "full" mode only: a token OUTSIDE the pre-beam candidates could have won (fewer live candidates than the beam) -
            # the search runs again on the tensor expressions over all V tokens
            keep, self.candidate_select = self.candidate_select, False
            try:
                return self._forward_batch(xs, maxlenratio)
            finally:
                self.candidate_select = keep

    def _forward_batch(self, xs, maxlenratio):
        if self.graph_steps:
            out = self._forward_batch_graphed(xs, maxlenratio)
            if out is not None:
                return out
        Ts = [int(x.shape[0]) for x in xs]
        maxlens = [T if maxlenratio == 0 else max(1, int(maxlenratio * T)) for T in Ts]
        with torch.no_grad():
            C_ = self._batch_consts(xs, Ts, maxlens, max(Ts), max(maxlens) + 2, always_mask=False)
            S = self._batch_state0(C_)
            run = _BatchLog(self, C_["B"], maxlens, maxlenratio, C_["allk"])
            for i in range(max(maxlens)):
                S, rec = self._batch_step(i, C_, S)
                if run.add(rec, last=(i == max(maxlens) - 1)):
                    break
        return run.results()

    # ---- one batched search in pieces: per-search constants, the state a step carries, the step -----------------------------------
    def _batch_consts(self, xs, Ts, maxlens, Tpad, W, always_mask):
        """tensors that stay the same over the steps of a search (Tpad >= max(Ts): frames the memory is padded to; W = width of
        the prefix buffer)"""
        B, V, beam, dev = len(xs), self.n_vocab, self.beam_size, xs[0].device
        n = B * beam
        names = list(self.full_scorers.keys())
        pname = next(iter(self.part_scorers), None)
        ctc = self.part_scorers[pname] if pname is not None else None
        xpad = torch.zeros(B, Tpad, xs[0].shape[-1], device=dev, dtype=xs[0].dtype)
        for b, x in enumerate(xs):
            xpad[b, :x.shape[0]] = x
        lens_d = torch.tensor(Ts, dtype=torch.int32, device=dev)
        C_ = dict(B=B, V=V, beam=beam, n=n, dev=dev, W=W, Tpad=Tpad, names=names, pname=pname, ctc=ctc,
                  allk=names + ([pname] if pname is not None else []), xpad=xpad, lens_d=lens_d,
                  uniform=(not always_mask) and all(T == Tpad for T in Ts))
        C_["xall"] = xpad.unsqueeze(1).expand(B, beam, Tpad, xpad.shape[-1]).reshape(n, Tpad, xpad.shape[-1])
        # [B, 1, Tpad], already in the kernels' uint8 (a bool mask was converted by every step's score_tree)
        C_["mem_mask1"] = (torch.arange(Tpad, device=dev)[None, :] < lens_d[:, None]).unsqueeze(1).to(torch.uint8)
        C_["sos32"] = torch.full((n,), self.sos, dtype=torch.int32, device=dev)
        C_["mem_mask"] = C_["mem_mask1"].unsqueeze(1).expand(B, beam, 1, Tpad).reshape(n, 1, Tpad)
        import inspect
        C_["masked"] = {k for k, d in self.full_scorers.items() if "memory_mask" in inspect.signature(d.score_tree).parameters}
        C_["base"] = (torch.arange(B, device=dev) * beam).view(B, 1)
        C_["maxlen_d"] = torch.tensor(maxlens, device=dev).view(B, 1)
        if ctc is not None:
            logp = ctc.ctc.log_softmax(xpad).contiguous()                                 # [B, Tpad, V]
            r0 = torch.full((B, Tpad, 2), -10000000000.0, device=dev, dtype=torch.float32)
            r0[:, :, 1] = torch.cumsum(logp[:, :, ctc.blank], 1)
            C_["logp"] = logp
            C_["c_r0"] = r0.unsqueeze(1).expand(B, beam, Tpad, 2).reshape(n, Tpad, 2).contiguous()
            C_["last_idx"] = (lens_d.long() - 1).view(B, 1).expand(B, beam).reshape(n).contiguous()    # last valid frame of each slot's utterance
        return C_

    def _batch_state0(self, C_):
        B, beam, n, dev = C_["B"], C_["beam"], C_["n"], C_["dev"]
        for d in self.full_scorers.values():         # a new search (scorers that keep per-search tensors drop them)
            if hasattr(d, "batch_init_state"):
                d.batch_init_state(C_["xpad"])
        yseq = torch.full((n, C_["W"]), self.eos, dtype=torch.int64, device=dev)
        yseq[:, 0] = self.sos
        hyp = torch.full((B, beam), -float("inf"), device=dev, dtype=torch.float32)
        hyp[:, 0] = 0.0
        S = dict(yseq=yseq, hyp=hyp.view(-1), sc=torch.zeros(len(C_["allk"]), n, device=dev, dtype=torch.float32),     # rows as allk
                 trees={k: None for k in C_["names"]})
        if C_["ctc"] is not None:
            S["c_s"] = torch.zeros(n, device=dev, dtype=torch.float32)
            S["c_r"] = C_["c_r0"]
        return S

    def _batch_step(self, i, C_, S, dyn=None):
        """step i of a batched search: state S -> (next state, log row [n, 3 + scorers + W]); no host synchronisation.
        dyn = dict(step=int32 device scalar, step_out=..., ring=[R, n, 3 + scorers + W]): the step index is READ FROM THE DEVICE by
        the kernels (i is ignored), the log row goes into slot step % R of the ring - a capture of this call serves every step >= 1
        (_forward_batch_graphed, graph_one).  Only the candidate-selection path with scorers that take tree["dyn"] runs that way:
        anything else raises _NoDynStep and the caller keeps one graph per step."""
        from .. import ops
        B, V, beam, n, dev = C_["B"], C_["V"], C_["beam"], C_["n"], C_["dev"]
        names, pname, ctc, allk = C_["names"], C_["pname"], C_["ctc"], C_["allk"]
        NEG = -float("inf")
        L = i + 1
        yseq, hyp, trees = S["yseq"], S["hyp"], dict(S["trees"])
        ys = yseq[:, :L] if dyn is None else yseq
        sdev = dyn["step"] if dyn is not None else None
        if dyn is not None:
            if ctc is None or "c_r" not in S or "last32" not in S or "tok" not in S:
                raise _NoDynStep("state")
            for k in names:        # the scorers read the newest tokens and the position from device memory
                if isinstance(trees[k], dict):
                    trees[k] = dict(trees[k], dyn=(sdev, S["tok"]))
                elif not getattr(self.full_scorers[k], "stateless_tree", False):
                    raise _NoDynStep("scorer " + k)
        # CTC forward variables of the running hypotheses: ready (first step / the full-recursion path), or still to be made from the
        # previous step's selection - then on a second stream BESIDE the decoder stack below (eamd_ctc_prefix_state: ~160 us of
        # frame-by-frame recursion that nothing in this step needs before the candidates are scored)
        c_r_now, side = S.get("c_r"), None
        if ctc is not None and c_r_now is None:
            pend = S["c_pend"]
            c_r_now = torch.empty(n, C_["Tpad"], 2, device=dev, dtype=torch.float32)
            if self.ctc_side_stream is True or (self.ctc_side_stream == "capture" and torch.cuda.is_current_stream_capturing()
                                                and C_["Tpad"] > 2048):
                side = self._ctc_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    ops.ctc_prefix_state(C_["logp"], C_["lens_d"], beam, *pend, ctc.blank, out=c_r_now)
            else:
                ops.ctc_prefix_state(C_["logp"], C_["lens_d"], beam, *pend, ctc.blank, out=c_r_now)
        logps, newtrees = {}, {}
        for k, d in self.full_scorers.items():
            # scorers that take it get the memory of the B utterances, not of the B * beam slots (shared_memory_ok)
            mem, mm = (C_["xpad"], C_["mem_mask1"]) if getattr(d, "shared_memory_ok", False) else (C_["xall"], C_["mem_mask"])
            if not C_["uniform"] and k in C_["masked"]:
                logps[k], newtrees[k] = d.score_tree(ys, trees[k], mem, memory_mask=mm)
            else:
                logps[k], newtrees[k] = d.score_tree(ys, trees[k], mem)
        P = self.pre_beam_size
        full_fast = (self.partial_mode == "full" and self.step_kernel and P <= 63 and P + 1 >= beam and beam * (P + 1) <= 1023
                     and beam <= 64 and beam * V < 2 ** 31 - 1024
                     and self.ctc_psi_parallel and C_["Tpad"] <= 2048)
        if (ctc is not None and self.do_pre_beam and (self.partial_mode == "ids" or full_fast) and self.pre_beam_score_key == "full"
                and 1 <= len(names) <= 4 and V % 4 == 0 and beam * P <= 1024 and self.candidate_select
                and all(logps[k].dtype == torch.float32 and logps[k].is_contiguous() for k in names)):
            # BeamSearch with a pre-beam: the step's selection on the beam x P candidates (csrc/decode.hip: eamd_weighted_sum,
            # eamd_beam_select) - same scores in the same order of operations as the tensor expressions below, 12 launches fewer
            # BatchBeamSearch ("full": the partial scorer reports a whole [n, V] row - log-zero outside the pre-beam, <eos> always
            # scored; nothing is masked): the same selection on P + 1 candidates, the pre-beam and <eos>.  Every other token's
            # score is ~ -3e9 (weight x log-zero): it can only win where an utterance has fewer live candidates than `beam` -
            # the host sees that in the step log (a winner below -1e9) and repeats the search on the tensor expressions.
            if full_fast:
                pre, part_ids, cand32 = ops.weighted_topk_rows([logps[k] for k in names], [self.weights[k] for k in names], P,
                                                               extra=ctc.eos)
            elif self.step_kernel and P <= 64:      # the weighted sum is formed inside the pre-beam's top-k launch
                pre, part_ids, cand32 = ops.weighted_topk_rows([logps[k] for k in names], [self.weights[k] for k in names], P)
            else:
                pre = ops.weighted_sum([logps[k] for k in names], [self.weights[k] for k in names])
                _, part_ids, cand32 = ops.topk_rows(pre, P, idx32=True)
            # the newest token of every prefix as int32: the previous step's selection wrote it (eamd_beam_step), <sos> at step 0
            last = S["last32"] if "last32" in S else (C_["sos32"] if i == 0 else ys[:, -1].to(torch.int32).contiguous())
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
                side = None
            psi = ops.ctc_prefix_psi(C_["logp"], C_["lens_d"], beam, c_r_now, cand32, last, L - 1 if dyn is None else 0, ctc.blank, ctc.eos,
                                     olen_dev=sdev) if self.ctc_psi_parallel else None
            if dyn is not None and psi is None:
                raise _NoDynStep("CTC candidates")
            r_new = None
            if psi is None:       # more than 2048 frames: the full recursion for every candidate
                olen = torch.full((n,), L - 1, dtype=torch.int32, device=dev)
                psi, r_new = ops.ctc_prefix_score_batch(C_["logp"], C_["lens_d"], beam, c_r_now, cand32, last, olen, ctc.blank, ctc.eos)
            slot_done = None
            if self.step_kernel and beam <= 64 and beam * P <= 1023 and beam * V < 2 ** 31 and r_new is None:
                # one scorer with a slot table (the cached decoder): its re-ordering rides in the same launch
                tabled = [k for k in names if isinstance(newtrees[k], dict) and "slot" in newtrees[k] and "pos" in newtrees[k]]
                slot_in = newtrees[tabled[0]]["slot"] if len(tabled) == 1 else None
                res = ops.beam_step(
                    pre, part_ids, psi, S["c_s"], hyp, self.weights[pname], B, beam, L, i, self.eos, C_["maxlen_d"].view(-1),
                    S["sc"], [logps[k] for k in names], yseq,
                    dyn=(sdev, dyn["step_out"], dyn["ring"]) if dyn is not None else None, slot_in=slot_in)
                sc_new, yseq, hyp_new, hyp_i, tok_i, tok32, cs_new, rec = res[:8]
                if slot_in is not None:
                    slot_done = (tabled[0], res[8])
            elif dyn is not None:
                raise _NoDynStep("selection kernel")
            else:
                top_s, top_i, c_loc = ops.beam_select(pre, part_ids, psi, S["c_s"], hyp, self.weights[pname], B, beam)
                sc_new, yseq, hyp_new, hyp_i, tok_i, pos, rec = ops.beam_finish(
                    top_s.reshape(-1), top_i.reshape(-1), beam, V, L, i, self.eos, C_["maxlen_d"].view(-1),
                    S["sc"], [logps[k] for k in names], c_loc, False, part_ids, yseq)
                cs_new, tok32 = psi[hyp_i, pos], tok_i.to(torch.int32)
            for k in names:
                if slot_done is not None and k == slot_done[0]:
                    trees[k] = dict(newtrees[k], slot=slot_done[1])
                else:
                    trees[k] = self._reorder(self.full_scorers[k], newtrees[k], hyp_i)
            T_ = dict(sc=sc_new, trees=trees, yseq=yseq, hyp=hyp_new, c_s=cs_new, last32=tok32, tok=tok_i)
            if r_new is not None:
                T_["c_r"] = r_new[hyp_i, pos]
            elif dyn is not None or self.ctc_side_stream is False or (self.ctc_side_stream == "capture" and C_["Tpad"] <= 2048):
                # the survivors' forward variables right away (a parallel scan of a few us up to 2048 frames)
                T_["c_r"] = ops.ctc_prefix_state(C_["logp"], C_["lens_d"], beam, c_r_now, hyp_i, tok_i, last, L - 1 if dyn is None else 0,
                                                 hyp_new, ctc.blank, olen_dev=sdev)
            else:     # ... or at the start of the next step, on a second stream beside its decoder stack (see the top of this function)
                T_["c_pend"] = (c_r_now, hyp_i, tok_i, last, L - 1, hyp_new)
            return T_, rec
        if dyn is not None:
            raise _NoDynStep("tensor-expression path")
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
        weighted = torch.zeros(n, V, device=dev, dtype=torch.float32)
        for k in names:
            weighted += self.weights[k] * logps[k]
        part_ids = None
        if self.do_pre_beam:
            pre = weighted if self.pre_beam_score_key == "full" else logps[self.pre_beam_score_key]
            part_ids = ops.topk_rows(pre.contiguous(), self.pre_beam_size)[1]
        if ctc is not None:
            c_s, c_r = S["c_s"], c_r_now
            last = ys[:, -1].to(torch.int32).contiguous()
            olen = torch.full((n,), L - 1, dtype=torch.int32, device=dev)
            ids = part_ids if part_ids is not None else torch.arange(V, device=dev).unsqueeze(0).expand(n, V)
            psi, r_new = ops.ctc_prefix_score_batch(C_["logp"], C_["lens_d"], beam, c_r, ids.to(torch.int32).contiguous(), last, olen,
                                                    ctc.blank, ctc.eos)
            if self.partial_mode == "full":
                full = torch.full((n, V), -10000000000.0, device=dev, dtype=torch.float32)
                full.scatter_(1, ids.long(), psi)
                full[:, ctc.eos] = torch.logsumexp(c_r[torch.arange(n, device=dev), C_["last_idx"]], dim=-1)
                full[:, ctc.blank] = -10000000000.0
                c_local = full - c_s[:, None]
                weighted += self.weights[pname] * c_local
            else:
                c_local = psi - c_s[:, None]
                if part_ids is not None:
                    kept = torch.full_like(weighted, NEG)
                    kept.scatter_(1, ids, torch.gather(weighted, 1, ids) + self.weights[pname] * c_local)
                    weighted = kept
                else:
                    weighted += self.weights[pname] * c_local
        weighted += hyp[:, None]
        # one launch of eamd_topk_rows (value descending, ties by ascending index).  torch.topk takes its multi-block path for
        # these sizes: six launches and a sort - and zeroes its counters with memset nodes, which a captured graph replays
        # wrongly on this ROCm from the second launch on (espnet_amd/graphs.py; the round-3 step-graph fault)
        # ... in two stages: the best `beam` of an utterance's beam x V continuations lie among the best `beam` of each of its
        # slots (rows of V elements stay in registers; ONE workgroup walking beam x V ten times took 300 us)
        s1, i1 = ops.topk_rows(weighted, beam)                                              # [n, beam] per slot
        top_s, i2 = ops.topk_rows(s1.view(B, beam * beam), beam)                            # [B, beam] among beam x beam
        top_i = (i2 // beam) * V + i1.view(B, beam * beam).gather(1, i2)
        # which hypothesis / token each winner is, the scores carried along, prefixes, end tests and the log row: one launch
        c_loc = ids_pos = None
        if ctc is not None:
            c_loc = c_local.contiguous()
            ids_pos = part_ids if part_ids is not None else None          # no pre-beam: the candidates are 0 .. V-1 in order
        sc_new, yseq, hyp_new, hyp_i, tok_i, pos, rec = ops.beam_finish(
            top_s.reshape(-1).contiguous(), top_i.reshape(-1).contiguous(), beam, V, L, i, self.eos, C_["maxlen_d"].view(-1),
            S["sc"], [logps[k].contiguous() for k in names], c_loc, self.partial_mode == "full", ids_pos, yseq)
        for k in names:
            trees[k] = self._reorder(self.full_scorers[k], newtrees[k], hyp_i)
        T_ = dict(sc=sc_new, trees=trees, yseq=yseq, hyp=hyp_new)
        if ctc is not None:
            T_["c_s"] = full[hyp_i, tok_i] if self.partial_mode == "full" else psi[hyp_i, pos]
            T_["c_r"] = r_new[hyp_i, pos]
        return T_, rec

    # ---- hipGraph replay of the steps ------------------------------------------------------------------------------------------
    # An eager beam step is ~180 launches of 10-15 us of HOST time each: the device idles most of a step.  The step is a pure device
    # function of (step index, state, per-search constants), so step i of every search with the same (utterances, beam, padded
    # frames, prefix width) is ONE captured graph: the memory is padded to a multiple of `graph_frame_bucket` frames (padded frames
    # masked: source attention, CTC lengths), the constants of a search are copied into the buffers the graphs were captured on,
    # and graph i reads the state graph i - 1 left (its output tensors are the input tensors graph i was captured with).  The first
    # search of a signature runs eagerly (it also runs every lazy initialisation outside a capture), the second one captures each
    # step and replays it, later ones only replay; steps a signature has not reached before are captured when first needed.
    # Anything that cannot be captured (a scorer that synchronises or copies from the host inside score_tree) ends graph mode for
    # this object - the search then runs eagerly as before.
    graph_steps = os.environ.get("EAMD_BEAM_GRAPH_STEPS", "0") == "1"      # opt-in (attribute or environment): 33 -> 46 utt/s at config 2
    graph_one = True                  # steps >= 1 of a search replay ONE graph that reads the step index from the device (else one per step)
    graph_frame_bucket = 32
    graph_max_signatures = 16         # least recently used signatures (their two graphs and static buffers) are dropped beyond this

    def _forward_batch_graphed(self, xs, maxlenratio):
        from .. import graphs, ops
        B, beam = len(xs), self.beam_size
        Ts = [int(x.shape[0]) for x in xs]
        maxlens = [T if maxlenratio == 0 else max(1, int(maxlenratio * T)) for T in Ts]
        fb = self.graph_frame_bucket
        Tpad = (max(Ts) + fb - 1) // fb * fb
        W = (Tpad if maxlenratio == 0 else max(1, int(maxlenratio * Tpad))) + 2
        sig = (B, beam, Tpad, W, float(maxlenratio), str(xs[0].dtype), xs[0].device.index)
        if not hasattr(self, "_step_graphs"):
            self._step_graphs = {}
        G = self._step_graphs.pop(sig, None)
        if G is not None:
            self._step_graphs[sig] = G                         # most recently used last
        while len(self._step_graphs) >= self.graph_max_signatures and (G is None or len(self._step_graphs) > self.graph_max_signatures):
            self._step_graphs.pop(next(iter(self._step_graphs)))
        with torch.no_grad():
            C_new = self._batch_consts(xs, Ts, maxlens, Tpad, W, always_mask=True)
            if G is None:                       # first search of this signature: eager, and its tensors become the static buffers
                G = self._step_graphs[sig] = dict(C=C_new, S0=None, graphs={}, states={}, recs={}, memos=None, searches=0)
            else:
                base = {G["C"][k].untyped_storage().data_ptr() for k in ("xpad", "mem_mask1")}
                for k, v in C_new.items():      # same shapes by construction of the signature
                    if k in ("xpad", "lens_d", "maxlen_d", "logp", "c_r0", "last_idx", "mem_mask1"):
                        G["C"][k].copy_(v)
                    elif k in ("xall", "mem_mask") and G["C"][k].untyped_storage().data_ptr() not in base:
                        G["C"][k].copy_(v)      # per-slot copies (B > 1); with one utterance they are views of the buffers above
            C_ = G["C"]
            G["searches"] += 1
            run = _BatchLog(self, B, maxlens, maxlenratio, C_["allk"])
            if G["searches"] == 1:
                S = self._batch_state0(C_)
                for i in range(max(maxlens)):
                    S, rec = self._batch_step(i, C_, S)
                    if run.add(rec, last=(i == max(maxlens) - 1)):
                        break
                return run.results()
            try:
                if G["S0"] is None:
                    G["S0"] = self._batch_state0(C_)          # static initial state (re-initialised in place below)
                    G["states"][0] = G["S0"]
                else:
                    fresh = self._batch_state0(C_)            # also tells the scorers that a new search starts
                    G["S0"]["yseq"].copy_(fresh["yseq"]); G["S0"]["hyp"].copy_(fresh["hyp"])
                    G["S0"]["sc"].copy_(fresh["sc"])
                    if "c_s" in fresh:
                        G["S0"]["c_s"].copy_(fresh["c_s"])
                        if G["S0"]["c_r"] is not C_["c_r0"]:
                            G["S0"]["c_r"].copy_(C_["c_r0"])
                dry = os.environ.get("EAMD_STEP_GRAPH_DRY") == "1"       # diagnostic: the static buffers without capture / replay
                Sd = G["S0"]
                for i in range(max(maxlens)):
                    if dry:
                        Sd, rec = self._batch_step(i, C_, Sd)
                        if run.add(rec, last=(i == max(maxlens) - 1)):
                            break
                        continue
                    if i >= 1 and self.graph_one and G.get("dyn") is not False:
                        # every step >= 1: ONE graph (the step index lives on the device); its state is loaded from step 0's outputs
                        if G.get("dyn") is None:
                            G["dyn"] = self._dyn_capture(C_, G)
                        D = G["dyn"]
                        if D:
                            if i == 1:
                                ops.copy_jobs(D["load"])
                            D["graph"].replay()
                            if run.add(D["ring"][i % D["ring"].shape[0]], last=(i == max(maxlens) - 1)):
                                break
                            continue
                    g = G["graphs"].get(i)
                    if g is None:
                        if G["memos"] is not None:            # per-search tensors of the scorers live in the buffers of graph 0
                            for d, m in G["memos"]:
                                d._kv_memo = m
                        torch.cuda.synchronize()
                        g = graphs.new_graph()
                        with torch.cuda.graph(g):
                            S1, rec = self._batch_step(i, C_, G["states"][i])
                        graphs.audit(g, "beam step graph")      # a memset node (e.g. torch.topk's scratch) would replay wrongly
                        G["graphs"][i], G["states"][i + 1], G["recs"][i] = g, S1, rec
                        if i == 0:
                            G["memos"] = [(d, d._kv_memo) for d in self.full_scorers.values() if getattr(d, "_kv_memo", None) is not None]
                    g.replay()
                    if run.add(G["recs"][i], last=(i == max(maxlens) - 1)):
                        break
                return run.results()
            except Exception as e:  # noqa: BLE001 - a step that cannot be captured: eager searches from now on
                import logging
                logging.getLogger(__name__).warning("beam-search step graphs disabled: %s", str(e)[:200])
                torch.cuda.synchronize()
                self.graph_steps = False
                self._step_graphs = {}
                return None

    def _dyn_capture(self, C_, G):
        """the one graph of steps >= 1 (graph_one): static state buffers, a capture of _batch_step(dyn=...) followed by ONE launch that
        copies the step's outputs back over the state (eamd_copy_jobs), and the load list that fills the state from step 0's outputs.
        -> dict, or False when a step cannot read its index from the device (then: one graph per step, as before)"""
        from .. import graphs, ops
        S1 = G["states"].get(1)
        need = ("yseq", "hyp", "sc", "c_s", "last32", "tok", "c_r")
        if S1 is None or any(k not in S1 for k in need) or self.sync_every < 1:
            return False
        dev, n = C_["dev"], C_["n"]
        S = {k: torch.empty_like(S1[k].contiguous()) for k in need}
        S["trees"] = {k: (dict(t, slot=torch.empty_like(t["slot"])) if isinstance(t, dict) else t) for k, t in S1["trees"].items()}
        if any(isinstance(t, dict) and "slot" not in t for t in S1["trees"].values()):
            return False
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        step_out = torch.zeros(1, dtype=torch.int32, device=dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        ring = torch.zeros(self.sync_every, n, 3 + len(C_["allk"]) + C_["W"], device=dev, dtype=torch.float32)
        slots = [k for k, t in S["trees"].items() if isinstance(t, dict)]
        load = [(S[k], S1[k].contiguous()) for k in need] + [(S["trees"][k]["slot"], S1["trees"][k]["slot"]) for k in slots] + [(step, one)]
        if len(load) > 16 or any(not S1[k].is_contiguous() for k in need):
            return False
        ops.copy_jobs(load)
        torch.cuda.synchronize()
        g = graphs.new_graph()
        try:
            with torch.cuda.graph(g):
                T_, _rec = self._batch_step(1, C_, S, dyn=dict(step=step, step_out=step_out, ring=ring))
                ops.copy_jobs([(S[k], T_[k]) for k in need] + [(S["trees"][k]["slot"], T_["trees"][k]["slot"]) for k in slots]
                              + [(step, step_out)])
        except _NoDynStep as e:
            import logging
            logging.getLogger(__name__).info("beam search: one graph per step (%s)", e)
            torch.cuda.synchronize()
            return False
        graphs.audit(g, "beam step graph (all steps)")
        return dict(graph=g, S=S, ring=ring, load=load, step=step, keep=(step_out, one))

    @_inference_call
    def forward(self, x, maxlenratio=0.0, minlenratio=0.0):
        """x: (T, D) encoder output.  Returns the ended hypotheses, best first."""
        from .. import ops
        ops.zero_arena_off()
        if self._device_loop_ok(x):
            if minlenratio == 0.0:       # one utterance = a batch of one: the same step code (selection / bookkeeping kernels, graphs)
                return self.forward_batch([x], maxlenratio, minlenratio)[0]
            return self._forward_device(x, maxlenratio, minlenratio)
        T = x.shape[0]
        maxlen = T if maxlenratio == 0 else max(1, int(maxlenratio * T))
        with torch.no_grad():
            running = self.init_hyp(x)
            ended = []
            for i in range(maxlen):
                best = self.search(running, x)
                running = self.post_process(i, maxlen, maxlenratio, best, ended)
                if maxlenratio == 0.0 and end_detect([h.asdict() for h in ended], i):
                    break
                if len(running) == 0:
                    break
        nbest = sorted(ended, key=lambda h: float(h.score), reverse=True)
        if len(nbest) == 0:      # beam_search.py:375-384
            return [] if minlenratio < 0.1 else self.forward(x, maxlenratio, max(0.0, minlenratio - 0.1))
        return nbest


def recognize_beam(model, enc_output, recog_args, char_list=None, rnnlm=None):
    """E2E.recognize for ctc_weight < 1 (reference: e2e_asr_transformer.py:286-477 / asr/pytorch_backend/recog.py).
    rnnlm: any BatchScorerInterface language model, fused with weight recog_args.lm_weight.
    Returns [{"score": float, "yseq": [int]}] n-best, yseq starts with <sos> and ends with <eos>."""
    ctc_weight = float(getattr(recog_args, "ctc_weight", 0.0))
    if model.ctc is None:
        ctc_weight = 0.0
    lm_weight = float(getattr(recog_args, "lm_weight", 0.0)) if rnnlm is not None else 0.0
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, lm=lm_weight,
                   length_bonus=float(getattr(recog_args, "penalty", 0.0)))
    scorers = dict(decoder=model.decoder, ctc=CTCPrefixScorer(model.ctc, model.eos) if ctc_weight > 0 else None,
                   lm=rnnlm, length_bonus=LengthBonus(model.odim))
    bs = BeamSearch(scorers, weights, int(recog_args.beam_size), model.odim, model.sos, model.eos,
                    pre_beam_score_key=None if ctc_weight == 1.0 else "full")
    hyps = bs(enc_output, float(getattr(recog_args, "maxlenratio", 0.0)), float(getattr(recog_args, "minlenratio", 0.0)))
    nbest = int(getattr(recog_args, "nbest", 1))
    return [{"score": float(h.score), "yseq": [int(t) for t in h.yseq], "scores": h.scores} for h in hyps[:nbest]]


class _BatchLog:
    """host side of a batched search: collects the step log, fetches it once per `sync_every` steps, replays the reference's
    ended-hypothesis bookkeeping and end detection per utterance (beam_search.py:404-458), keeps the n-best lists"""

    def __init__(self, bs, B, maxlens, maxlenratio, allk):
        self.bs, self.B, self.maxlens, self.maxlenratio, self.allk = bs, B, maxlens, maxlenratio, allk
        self.ended = [[] for _ in range(B)]
        self.ended_sl = [[] for _ in range(B)]            # (score, length) of the ended hypotheses: what end detection reads
        self.stopped = [False] * B
        self.pending = []

    def add(self, rec, last):
        """-> True when every utterance has finished"""
        self.pending.append(rec)
        if len(self.pending) >= self.bs.sync_every or last:
            return self.flush()
        return False

    def flush(self):
        """the fetched log rows, vectorised over slots: only the slots that END in a step are walked in Python (with the whole
        [n, 3 + scorers + W] row turned into Python floats, a fetch of 8 steps x 320 slots took 2 - 20 ms of host time in
        which the device idled)"""
        import numpy as np
        bs, B, beam, allk = self.bs, self.B, self.bs.beam_size, self.allk
        if not self.pending:
            return all(self.stopped)
        host = torch.stack(self.pending).cpu().numpy()           # [steps, n, 3 + len(allk) + W]
        self.pending = []
        na = len(allk)
        maxl = np.asarray(self.maxlens)
        # everything that does not depend on which utterances have stopped, for all fetched steps at once (a numpy call costs a
        # microsecond or two: per step and per quantity they were most of a fetch)
        S_ = host.shape[0]
        steps = host[:, 0, 0].astype(np.int64)
        ts_all = host[:, :, 1].reshape(S_, B, beam)
        tok_all = host[:, :, 2].reshape(S_, B, beam)
        fin_all = np.isfinite(ts_all)
        cap_all = (maxl[None, :] - 1 == steps[:, None])                        # [steps, B]
        ends_all = fin_all & ((tok_all == bs.eos) | cap_all[:, :, None])
        alive_all = (fin_all & ~ends_all).sum(2)                               # [steps, B]
        es, eb, ej = np.nonzero(ends_all)
        guard = getattr(bs, "partial_mode", "ids") == "full" and getattr(bs, "candidate_select", False)
        low_all = (fin_all & (ts_all < -1e9)).any(2) if guard else None        # [steps, B]
        k = 0
        for si in range(S_):
            i = int(steps[si])
            L = i + 2
            row = host[si]
            if guard and any(low_all[si, b] and not self.stopped[b] for b in range(B)):
                raise _OutsideCandidates()
            while k < len(es) and es[k] == si:
                b, j = int(eb[k]), int(ej[k])
                k += 1
                if self.stopped[b]:
                    continue
                slot = row[b * beam + j]
                top_s = float(slot[1])
                seq = slot[3 + na: 3 + na + L].astype(np.int64).tolist()
                if i == self.maxlens[b] - 1:
                    seq.append(bs.eos)
                scores = {kk: float(slot[3 + q]) for q, kk in enumerate(allk)}
                if bs.apply_final_score:
                    for kk, d in chain(bs.full_scorers.items(), bs.part_scorers.items()):
                        f = float(d.final_tree(None)) if hasattr(d, "final_tree") else float(d.final_score(None))
                        scores[kk] += f
                        top_s += bs.weights[kk] * f
                self.ended[b].append(Hypothesis(yseq=torch.tensor(seq, dtype=torch.int64), score=top_s, scores=scores, states={}))
                self.ended_sl[b].append((top_s, len(seq)))
            for b in range(B):
                if self.stopped[b]:
                    continue
                if (self.maxlenratio == 0.0 and _end_detect_sl(self.ended_sl[b], i)) or alive_all[si, b] == 0 or i == self.maxlens[b] - 1:
                    self.stopped[b] = True
            if all(self.stopped):
                return True
        return False

    def results(self):
        return [sorted(self.ended[b], key=lambda h: float(h.score), reverse=True) for b in range(self.B)]
