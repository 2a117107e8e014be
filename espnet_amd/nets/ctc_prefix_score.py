"""CTC prefix scorer for joint CTC/attention decoding on the HIP kernels.

reference: espnet/nets/ctc_prefix_score.py:224-310 (CTCPrefixScore, per hypothesis, numpy) and :12-221
(CTCPrefixScoreTH, all hypotheses at once), espnet/nets/scorers/ctc.py:11-127 (CTCPrefixScorer),
espnet/nets/scorer_interface.py (partial-scorer protocol).

Both of the reference's implementations map onto ONE launch of eamd_ctc_prefix_score per beam step
(thread = (hypothesis, candidate), scan over frames).  State of one hypothesis = (prefix score s, r [T,2]),
both on the device.  The two public entry points differ only in what the reference returns around the
recursion:
  score_partial / score_partial_batch   scores of exactly the requested ids          (BeamSearch)
  batch_score_partial                   a full (n, V) matrix: log-zero outside the requested ids,
                                        <eos> always scored, blank excluded            (BatchBeamSearch)
"""
import torch

from .. import ops
from .scorer_interface import BatchPartialScorerInterface, BatchScorerInterface

LOGZERO = -10000000000.0


class CTCPrefixScorer(BatchPartialScorerInterface):
    def __init__(self, ctc, eos):
        self.ctc = ctc
        self.eos = eos
        self.blank = 0
        self.logp = None

    # ---- ScorerInterface ----------------------------------------------------------------------
    def init_state(self, x):
        """x: (T, D) encoder output -> (0.0, r0) with r0[t] = (logzero, sum_{s<=t} logp[s, blank])"""
        with torch.no_grad():
            self.logp = self.ctc.log_softmax(x.unsqueeze(0)).squeeze(0).contiguous()
        T = self.logp.shape[0]
        r = torch.full((T, 2), LOGZERO, device=x.device, dtype=torch.float32)
        r[:, 1] = torch.cumsum(self.logp[:, self.blank], 0)   # prefix sums: host-issued bookkeeping, T values
        return torch.zeros((), device=x.device), r

    def batch_init_state(self, x):
        return self.init_state(x)

    def select_state(self, state, i, new_id=None):
        """state as returned by the scoring calls -> state of hypothesis i extended by token new_id"""
        if state is None:
            return None
        if len(state) == 2:            # (psi [n,P], r_new [n,P,T,2]): i is (hyp, position in ids)
            sc, st = state
            return sc[i], st[i]
        psi, r_new, idmap = state      # full-matrix form: look the token up in the id map
        j = int(idmap[i, new_id]) if new_id is not None else 0
        return psi[i, new_id], r_new[i, max(j, 0)]

    def final_score(self, state):
        return 0.0

    # ---- the recursion: every running hypothesis in one launch -----------------------------------
    def _launch(self, ys, ids, states):
        dev = self.logp.device
        ys = torch.as_tensor(ys)
        n, L = ys.shape
        r_prev = torch.stack([s[1] for s in states])
        s_prev = torch.stack([torch.as_tensor(s[0], dtype=torch.float32, device=dev).reshape(()) for s in states])
        last = ys[:, -1].to(device=dev, dtype=torch.int32).contiguous()
        olen = torch.full((n,), L - 1, dtype=torch.int32, device=dev)
        psi, r_new = ops.ctc_prefix_score(self.logp, r_prev, ids.to(torch.int32).contiguous(), last, olen,
                                          self.blank, self.eos)
        return psi, r_new, s_prev, r_prev

    def score_partial_batch(self, ys, ids, states, x=None):
        """ys [n, L] prefixes (all of one length), ids [n, P] -> (scores of ids [n, P], (psi, r_new))"""
        psi, r_new, s_prev, _ = self._launch(ys, ids, states)
        return psi - s_prev[:, None], (psi, r_new)

    def score_partial(self, y, ids, state, x):
        """single-hypothesis form of the reference interface (scorers/ctc.py:65-83)"""
        d, (psi, r_new) = self.score_partial_batch(torch.as_tensor(y).view(1, -1), ids.view(1, -1), [state])
        return d[0], (psi[0], r_new[0])

    def batch_score_partial(self, ys, ids, states, x=None):
        """reference: scorers/ctc.py:97-127 -> CTCPrefixScoreTH.__call__ (ctc_prefix_score.py:70-188).
        ids None = score the whole vocabulary."""
        dev = self.logp.device
        n = len(states)
        V = self.logp.shape[1]
        if ids is None:
            ids = torch.arange(V, device=dev, dtype=torch.int32).unsqueeze(0).expand(n, V)
        psi, r_new, s_prev, r_prev = self._launch(ys, ids, states)
        full = torch.full((n, V), LOGZERO, device=dev, dtype=torch.float32)
        full.scatter_(1, ids.long(), psi)
        full[:, self.eos] = torch.logsumexp(r_prev[:, -1, :], dim=-1)        # :182-183, also outside the pre-beam
        full[:, self.blank] = LOGZERO                                        # :186
        idmap = torch.full((n, V), -1, dtype=torch.int64, device=dev)
        idmap.scatter_(1, ids.long(), torch.arange(ids.shape[1], device=dev).expand(n, -1))
        return full - s_prev[:, None], (full, r_new, idmap.cpu())


class LengthBonus(BatchScorerInterface):
    """reference: espnet/nets/scorers/length_bonus.py:11-61 (+1 per emitted token)"""

    stateless_tree = True          # score_tree needs neither the step index nor a state (BeamSearch's one-graph steps)

    def __init__(self, n_vocab):
        self.n = n_vocab

    def init_state(self, x):
        return None

    def batch_init_state(self, x):
        return None

    def select_state(self, state, i, new_id=None):
        return None

    def final_score(self, state):
        return 0.0

    def score(self, y, state, x):
        return torch.ones(self.n, device=x.device, dtype=x.dtype), None

    def batch_score(self, ys, states, xs):
        return torch.ones(len(ys), self.n, device=xs.device, dtype=xs.dtype), None

    def score_tree(self, ys, tree, xs):
        return torch.ones(len(ys), self.n, device=xs.device, dtype=xs.dtype), None

    def final_tree(self, tree):
        return 0.0
