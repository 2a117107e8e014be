"""Scorer protocol of the joint decoding loop (reference: espnet/nets/scorer_interface.py:11-188).

Same class names and method signatures.  When the reference package is importable its classes are used
as the bases, so `isinstance(decoder, ScorerInterface)` inside the reference's own BeamSearch holds for
our modules; otherwise structurally identical stand-alone bases are defined here.
"""
try:  # pragma: no cover - only when the reference is installed next to us
    from espnet.nets.scorer_interface import (BatchPartialScorerInterface, BatchScorerInterface,  # type: ignore
                                              PartialScorerInterface, ScorerInterface)
except Exception:  # noqa: BLE001

    class ScorerInterface:
        """full-vocabulary scorer: score(y, state, x) -> (scores [V], new state)"""

        def init_state(self, x):
            return None

        def select_state(self, state, i, new_id=None):
            return None if state is None else state[i]

        def score(self, y, state, x):
            raise NotImplementedError

        def final_score(self, state):
            return 0.0

    class BatchScorerInterface(ScorerInterface):
        """batch_score(ys [n, L], states list[n], xs [n, T, D]) -> (scores [n, V], states list[n])"""

        def batch_init_state(self, x):
            return self.init_state(x)

        def batch_score(self, ys, states, xs):
            scores, outstates = [], []
            for y, state, x in zip(ys, states, xs):
                s, o = self.score(y, state, x)
                scores.append(s)
                outstates.append(o)
            import torch
            return torch.stack(scores), outstates

    class PartialScorerInterface(ScorerInterface):
        """score_partial(y, next_tokens, state, x) -> (scores [len(next_tokens)], new state)"""

        def score_partial(self, y, next_tokens, state, x):
            raise NotImplementedError

    class BatchPartialScorerInterface(BatchScorerInterface, PartialScorerInterface):
        """batch_score_partial(ys, next_tokens [n, P], states, xs) -> (scores [n, V], new states)"""

        def batch_score_partial(self, ys, next_tokens, states, xs):
            raise NotImplementedError
