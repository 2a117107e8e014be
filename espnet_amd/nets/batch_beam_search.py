"""Vectorised beam search.  reference: espnet/nets/batch_beam_search.py:31-348 (BatchBeamSearch).

The search loop of `BeamSearch` here is already batched over the running hypotheses; what this class keeps
of the reference's BatchBeamSearch is its arithmetic, which differs from BeamSearch's in three places:
  * partial scorers return full (n, V) matrices (CTCPrefixScoreTH): tokens outside the pre-beam keep their
    full score plus weight * (log-zero - s_prev) instead of being dropped, <eos> is always CTC-scored and
    blank is excluded (ctc_prefix_score.py:165-188; batch_beam_search.py:221-231);
  * the top-`beam` is taken over the flattened (n * V) matrix (batch_beam_search.py:86-110);
  * post_process applies no `final_score` (batch_beam_search.py:288-348).
"""
from .beam_search import BeamSearch, Hypothesis  # noqa: F401


class BatchBeamSearch(BeamSearch):
    partial_mode = "full"
    apply_final_score = False
