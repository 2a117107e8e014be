"""CTC prefix scorer for joint CTC/attention decoding on the HIP kernels.

reference: espnet/nets/ctc_prefix_score.py:224-310 (CTCPrefixScore), espnet/nets/scorers/ctc.py:11-127
(CTCPrefixScorer), espnet/nets/scorer_interface.py (partial-scorer protocol).
State of one hypothesis = (previous prefix score, r [T,2] device tensor).  All hypotheses of a beam step
are scored by ONE launch of eamd_ctc_prefix_score (thread = (hypothesis, candidate), scan over frames).
"""
import torch

from .. import ops

LOGZERO = -10000000000.0


class CTCPrefixScorer:
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
        return 0.0, r

    def select_state(self, state, i, new_id=None):
        sc, st = state
        return float(sc[i]), st[i]

    def final_score(self, state):
        return 0.0

    # ---- batched partial scoring: every running hypothesis in one launch ------------------------
    def batch_score_partial(self, yseqs, cand, states):
        """yseqs: list of token lists; cand: int32 [nhyp, ncand] device; states: list of (prev_score, r)
        returns (delta scores [nhyp, ncand] device, (psi [nhyp,ncand] host list, r_new [nhyp,ncand,T,2]))"""
        dev = cand.device
        r_prev = torch.stack([s[1] for s in states])
        last = torch.tensor([int(y[-1]) for y in yseqs], dtype=torch.int32).to(dev)
        olen = torch.tensor([len(y) - 1 for y in yseqs], dtype=torch.int32).to(dev)
        psi, r_new = ops.ctc_prefix_score(self.logp, r_prev, cand, last, olen, self.blank, self.eos)
        prev = torch.tensor([float(s[0]) for s in states], dtype=torch.float32).to(dev)
        return psi - prev[:, None], (psi, r_new)

    def score_partial(self, y, ids, state, x):
        """single-hypothesis form of the reference interface"""
        d, (psi, r_new) = self.batch_score_partial([[int(v) for v in y]], ids.to(torch.int32).view(1, -1), [state])
        return d[0], (psi[0], r_new[0])


class LengthBonus:
    """reference: espnet/nets/scorers/length_bonus.py:11-61 (+1 per emitted token)"""

    def __init__(self, n_vocab):
        self.n = n_vocab

    def init_state(self, x):
        return None

    def final_score(self, state):
        return 0.0
