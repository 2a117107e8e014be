"""Speech2Text: encoder + joint CTC/attention(/LM) beam search for one utterance.

reference: espnet2/bin/asr_inference.py:36-211.  The reference builds its models from a training config
through `ASRTask.build_model_from_file` / `LMTask.build_model_from_file`; that task registry is control
plane (SURVEY.md §8 out of scope), so this class takes the built modules instead - an `ESPnetASRModel`
(ours or the reference's, holding our encoder / decoder / CTC) and optionally a language model.  Everything
after model construction follows the reference: scorer and weight dictionaries, BatchBeamSearch when every
full scorer is a BatchScorerInterface, `__call__(speech) -> [(text, token, token_int, hyp)]`.
"""
import numpy as np
import torch

from .. import ops

from ..nets.batch_beam_search import BatchBeamSearch
from ..nets.beam_search import BeamSearch, Hypothesis
from ..nets.ctc_prefix_score import CTCPrefixScorer, LengthBonus
from ..nets.scorer_interface import BatchScorerInterface


class TokenIDConverter:
    """reference: espnet2/text/token_id_converter.py (list-backed subset)"""

    def __init__(self, token_list, unk_symbol="<unk>"):
        self.token_list = list(token_list)
        self.token2id = {t: i for i, t in enumerate(self.token_list)}
        self.unk_id = self.token2id.get(unk_symbol, None)

    def ids2tokens(self, integers):
        return [self.token_list[i] for i in integers]

    def tokens2ids(self, tokens):
        return [self.token2id.get(t, self.unk_id) for t in tokens]


class CharTokenizer:
    """reference: espnet2/text/char_tokenizer.py (tokens2text: join, <space> -> ' ')"""

    def __init__(self, space_symbol="<space>"):
        self.space_symbol = space_symbol

    def tokens2text(self, tokens):
        return "".join(" " if t == self.space_symbol else t for t in tokens)


class Speech2Text:
    def __init__(self, asr_model, lm=None, token_list=None, tokenizer=None, device="cuda", maxlenratio=0.0,
                 minlenratio=0.0, batch_size=1, beam_size=20, ctc_weight=0.5, lm_weight=1.0, penalty=0.0, nbest=1,
                 graph_steps=False):
        asr_model.to(device).eval()
        token_list = token_list if token_list is not None else getattr(asr_model, "token_list", None)
        vocab = len(token_list) if token_list is not None else asr_model.vocab_size
        scorers = dict(decoder=asr_model.decoder, ctc=CTCPrefixScorer(ctc=asr_model.ctc, eos=asr_model.eos),
                       length_bonus=LengthBonus(vocab))
        if lm is not None:
            scorers["lm"] = lm.to(device).eval()
        weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, lm=lm_weight, length_bonus=penalty)
        beam_search = BeamSearch(beam_size=beam_size, weights=weights, scorers=scorers, sos=asr_model.sos,
                                 eos=asr_model.eos, vocab_size=vocab, token_list=token_list,
                                 pre_beam_score_key=None if ctc_weight == 1.0 else "full")
        if batch_size == 1 and all(isinstance(v, BatchScorerInterface) for v in beam_search.full_scorers.values()):
            beam_search.__class__ = BatchBeamSearch          # asr_inference.py:108-118
        beam_search.to(device).eval()
        # graph_steps (not in the reference): the steps of a search replayed as hipGraphs, see nets.beam_search.BeamSearch.graph_steps
        beam_search.graph_steps = bool(graph_steps)
        self.asr_model, self.beam_search = asr_model, beam_search
        self.converter = TokenIDConverter(token_list) if token_list is not None else None
        self.tokenizer = tokenizer
        self.maxlenratio, self.minlenratio, self.device, self.nbest = maxlenratio, minlenratio, device, nbest

    @torch.no_grad()
    @ops.inference_call
    def __call__(self, speech):
        """speech: (Nsamples,) waveform or (T, F) features, as the model's frontend expects"""
        if isinstance(speech, np.ndarray):
            speech = torch.tensor(speech)
        speech = speech.unsqueeze(0).to(torch.float32).to(self.device)
        lengths = torch.full([1], speech.size(1), dtype=torch.long)
        enc, _ = self.asr_model.encode(speech=speech, speech_lengths=lengths)
        assert len(enc) == 1, len(enc)
        nbest_hyps = self.beam_search(x=enc[0], maxlenratio=self.maxlenratio, minlenratio=self.minlenratio)
        results = []
        for hyp in nbest_hyps[: self.nbest]:
            assert isinstance(hyp, Hypothesis), type(hyp)
            token_int = [t for t in hyp.yseq[1:-1].tolist() if t != 0]     # drop sos/eos and the blank id 0
            token = self.converter.ids2tokens(token_int) if self.converter is not None else None
            text = self.tokenizer.tokens2text(token) if (self.tokenizer is not None and token is not None) else None
            results.append((text, token, token_int, hyp))
        return results
