"""espnet2 plug-in surface: encoder / decoder / CTC classes with the AbsEncoder / AbsDecoder call
signatures and `ESPnetASRModel.forward(speech, speech_lengths, text, text_lengths) -> (loss, stats, weight)`.

reference: espnet2/asr/encoder/conformer_encoder.py:39-255, espnet2/asr/encoder/transformer_encoder.py:28-175,
espnet2/asr/decoder/transformer_decoder.py:31-275, espnet2/asr/ctc.py:6-111,
espnet2/asr/espnet_model.py:35-290, espnet2/train/abs_espnet_model.py,
registries: espnet2/tasks/asr.py:77-101 (`encoder_choices`, `decoder_choices`).
All arithmetic is the espnet1-level block set in espnet_amd.nets.modules (one implementation, two surfaces,
exactly as the reference does).
"""
import torch

from .. import functional as F_
from .. import ops
from ..nets import modules as M
from ..nets.scorer_interface import BatchScorerInterface

try:  # pragma: no cover - only when the reference package is importable
    from espnet2.asr.decoder.abs_decoder import AbsDecoder  # type: ignore
    from espnet2.asr.encoder.abs_encoder import AbsEncoder  # type: ignore
    from espnet2.train.abs_espnet_model import AbsESPnetModel  # type: ignore
except Exception:  # noqa: BLE001
    AbsEncoder = AbsDecoder = AbsESPnetModel = torch.nn.Module


def _lens(x):
    return [int(v) for v in (x.tolist() if isinstance(x, torch.Tensor) else x)]


class ConformerEncoder(AbsEncoder):
    """reference: espnet2/asr/encoder/conformer_encoder.py:39-255 (same defaults: rel_pos, rel_selfattn,
    swish, macaron False, cnn module True with kernel 31)."""

    def __init__(self, input_size, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 normalize_before=True, concat_after=False, positionwise_layer_type="linear",
                 positionwise_conv_kernel_size=3, macaron_style=False, pos_enc_layer_type="rel_pos",
                 selfattention_layer_type="rel_selfattn", activation_type="swish", use_cnn_module=True,
                 cnn_module_kernel=31, padding_idx=-1):
        super().__init__()
        self._output_size = output_size
        enc = M.ConformerEncoder(
            idim=input_size, attention_dim=output_size, attention_heads=attention_heads, linear_units=linear_units,
            num_blocks=num_blocks, dropout_rate=dropout_rate, positional_dropout_rate=positional_dropout_rate,
            attention_dropout_rate=attention_dropout_rate, input_layer=input_layer, normalize_before=normalize_before,
            concat_after=concat_after, positionwise_layer_type=positionwise_layer_type,
            positionwise_conv_kernel_size=positionwise_conv_kernel_size, macaron_style=macaron_style,
            pos_enc_layer_type=pos_enc_layer_type, selfattention_layer_type=selfattention_layer_type,
            activation_type=activation_type, use_cnn_module=use_cnn_module, cnn_module_kernel=cnn_module_kernel,
            padding_idx=padding_idx)
        # same attribute names as the reference class => identical state_dict keys
        self.embed, self.encoders, self.after_norm = enc.embed, enc.encoders, enc.after_norm
        self.normalize_before = normalize_before

    def output_size(self):
        return self._output_size

    def forward(self, xs_pad, ilens, prev_states=None):
        il = _lens(ilens)
        self._tin = xs_pad.size(1)
        masks = M.make_non_pad_mask(il, xs_pad.size(1))[:, None, :].to(xs_pad.device).to(torch.uint8)
        xs_pad, masks = self.embed(xs_pad, masks)
        xs_pad, masks = self.encoders(xs_pad, masks)
        if isinstance(xs_pad, tuple):
            xs_pad = xs_pad[0]
        if self.normalize_before:
            xs_pad = self.after_norm(xs_pad)
        olens = torch.tensor(M.embed_output_lengths(self.embed, il, self._tin), dtype=torch.int64).to(xs_pad.device)
        return xs_pad, olens, None


class TransformerEncoder(AbsEncoder):
    """reference: espnet2/asr/encoder/transformer_encoder.py:28-175"""

    def __init__(self, input_size, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d",
                 pos_enc_class=M.PositionalEncoding, normalize_before=True, concat_after=False,
                 positionwise_layer_type="linear", positionwise_conv_kernel_size=1, padding_idx=-1):
        super().__init__()
        self._output_size = output_size
        enc = M.TransformerEncoder(
            idim=input_size, attention_dim=output_size, attention_heads=attention_heads, linear_units=linear_units,
            num_blocks=num_blocks, dropout_rate=dropout_rate, positional_dropout_rate=positional_dropout_rate,
            attention_dropout_rate=attention_dropout_rate, input_layer=input_layer, pos_enc_class=pos_enc_class,
            normalize_before=normalize_before, concat_after=concat_after,
            positionwise_layer_type=positionwise_layer_type,
            positionwise_conv_kernel_size=positionwise_conv_kernel_size, padding_idx=padding_idx)
        self.embed, self.encoders, self.after_norm = enc.embed, enc.encoders, enc.after_norm
        self.normalize_before = normalize_before

    def output_size(self):
        return self._output_size

    def forward(self, xs_pad, ilens, prev_states=None):
        il = _lens(ilens)
        self._tin = xs_pad.size(1)
        masks = M.make_non_pad_mask(il, xs_pad.size(1))[:, None, :].to(xs_pad.device).to(torch.uint8)
        xs_pad, masks = self.embed(xs_pad, masks)
        xs_pad, masks = self.encoders(xs_pad, masks)
        if self.normalize_before:
            xs_pad = self.after_norm(xs_pad)
        olens = torch.tensor(M.embed_output_lengths(self.embed, il, self._tin), dtype=torch.int64).to(xs_pad.device)
        return xs_pad, olens, None


class TransformerDecoder(AbsDecoder, BatchScorerInterface):
    """reference: espnet2/asr/decoder/transformer_decoder.py:31-275 (BaseTransformerDecoder + TransformerDecoder)"""

    def __init__(self, vocab_size, encoder_output_size, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, self_attention_dropout_rate=0.0,
                 src_attention_dropout_rate=0.0, input_layer="embed", use_output_layer=True,
                 pos_enc_class=M.PositionalEncoding, normalize_before=True, concat_after=False):
        super().__init__()
        dec = M.Decoder(odim=vocab_size, attention_dim=encoder_output_size, attention_heads=attention_heads,
                        linear_units=linear_units, num_blocks=num_blocks, dropout_rate=dropout_rate,
                        positional_dropout_rate=positional_dropout_rate,
                        self_attention_dropout_rate=self_attention_dropout_rate,
                        src_attention_dropout_rate=src_attention_dropout_rate, input_layer=input_layer,
                        use_output_layer=use_output_layer, pos_enc_class=pos_enc_class,
                        normalize_before=normalize_before, concat_after=concat_after)
        self._dec = [dec]   # not registered twice: the submodules below carry the parameters
        # registration order of the reference (BaseTransformerDecoder first, then the subclass' layers)
        self.embed, self.after_norm, self.output_layer = dec.embed, dec.after_norm, dec.output_layer
        self.decoders = dec.decoders
        self.normalize_before = normalize_before

    def forward(self, hs_pad, hlens, ys_in_pad, ys_in_lens):
        dec = self._dec[0]
        dev = hs_pad.device
        yl, hl = _lens(ys_in_lens), _lens(hlens)
        U = ys_in_pad.size(1)
        tgt_mask = (M.make_non_pad_mask(yl, U)[:, None, :] & M.subsequent_mask(U).unsqueeze(0)).to(torch.uint8)
        memory_mask = M.make_non_pad_mask(hl, hs_pad.size(1))[:, None, :].to(torch.uint8)
        x, _ = dec(ys_in_pad, tgt_mask.to(dev).contiguous(), hs_pad, memory_mask.to(dev).contiguous())
        olens = torch.tensor(yl, dtype=torch.int64).to(dev)
        return x, olens

    # scorer API (reference: transformer_decoder.py:150-245)
    def forward_one_step(self, tgt, tgt_mask, memory, cache=None):
        return self._dec[0].forward_one_step(tgt, tgt_mask, memory, cache=cache)

    def init_state(self, x):
        return None

    def score(self, ys, state, x):
        return self._dec[0].score(ys, state, x)

    def batch_score(self, ys, states, xs):
        return self._dec[0].batch_score(ys, states, xs)

    shared_memory_ok = True      # see nets.modules.Decoder: the searches hand over the memory of G utterances, not of G * beam rows

    def batch_init_state(self, x):
        return self._dec[0].batch_init_state(x)

    def score_tree(self, ys, tree, xs, memory_mask=None):
        return self._dec[0].score_tree(ys, tree, xs, memory_mask=memory_mask)

    def final_tree(self, tree):
        return 0.0


class CTC(torch.nn.Module):
    """reference: espnet2/asr/ctc.py:6-111 (forward takes ys_lens; builtin and warpctc compute the same
    sum_b nll / B)."""

    def __init__(self, odim, encoder_output_sizse, dropout_rate=0.0, ctc_type="builtin", reduce=True):
        super().__init__()
        self._ctc = [M.CTC(odim, encoder_output_sizse, dropout_rate, ctc_type=ctc_type, reduce=reduce)]
        self.ctc_lo = self._ctc[0].ctc_lo
        self.dropout_rate = dropout_rate
        self.ctc_type = ctc_type
        self.reduce = reduce

    def forward(self, hs_pad, hlens, ys_pad, ys_lens):
        # labels beyond ys_lens are ignored exactly as `ys_pad[i, :l]` does in the reference
        L = ys_pad.size(1)
        keep = torch.arange(L, device=ys_pad.device)[None, :] < torch.as_tensor(ys_lens, device=ys_pad.device)[:, None]
        ys = torch.where(keep, ys_pad, torch.full_like(ys_pad, -1))
        return self._ctc[0](hs_pad, hlens, ys.contiguous())

    def log_softmax(self, hs_pad):
        return self._ctc[0].log_softmax(hs_pad)

    def argmax(self, hs_pad):
        return self._ctc[0].argmax(hs_pad)


class ESPnetASRModel(AbsESPnetModel):
    """reference: espnet2/asr/espnet_model.py:35-290.  frontend = None (fbank features are the input) or
    espnet_amd.espnet2.DefaultFrontend (waveform input); specaug and normalize take the espnet_amd.espnet2.layers
    modules (SpecAug, GlobalMVN, UtteranceMVN)."""

    def __init__(self, vocab_size, token_list=None, frontend=None, specaug=None, normalize=None, encoder=None,
                 decoder=None, ctc=None, rnnt_decoder=None, ctc_weight=0.5, ignore_id=-1, lsm_weight=0.0,
                 length_normalized_loss=False, report_cer=False, report_wer=False, sym_space="<space>",
                 sym_blank="<blank>"):
        assert 0.0 <= ctc_weight <= 1.0, ctc_weight
        assert rnnt_decoder is None, "Not implemented"
        super().__init__()
        self.sos = vocab_size - 1
        self.eos = vocab_size - 1
        self.vocab_size = vocab_size
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.token_list = list(token_list) if token_list is not None else None
        self.frontend, self.specaug, self.normalize = frontend, specaug, normalize
        self.encoder = encoder
        self.decoder = decoder
        self.ctc = None if ctc_weight == 0.0 else ctc
        self.rnnt_decoder = None
        self.criterion_att = M.LabelSmoothingLoss(size=vocab_size, padding_idx=ignore_id, smoothing=lsm_weight,
                                                  normalize_length=length_normalized_loss)
        self.error_calculator = None

    def encode(self, speech, speech_lengths):
        """reference: espnet_model.py:178-213 (features = speech when frontend is None; SpecAug in training mode
        only, then the normalisation layer, :192-197)"""
        feats, feats_lengths = self._extract_feats(speech, speech_lengths)
        if self.specaug is not None and self.training:
            feats, feats_lengths = self.specaug(feats, feats_lengths)
        if self.normalize is not None:
            feats, feats_lengths = self.normalize(feats, feats_lengths)
        encoder_out, encoder_out_lens, _ = self.encoder(feats, feats_lengths)
        assert encoder_out.size(0) == speech.size(0)
        return encoder_out, encoder_out_lens

    def _extract_feats(self, speech, speech_lengths):
        """reference: espnet_model.py:214-233"""
        speech = speech[:, : int(max(_lens(speech_lengths)))]
        if self.frontend is not None:
            return self.frontend(speech.contiguous(), speech_lengths)
        return speech, speech_lengths

    def collect_feats(self, speech, speech_lengths, text, text_lengths):
        feats, feats_lengths = self._extract_feats(speech, speech_lengths)
        return {"feats": feats, "feats_lengths": feats_lengths}

    def forward(self, speech, speech_lengths, text, text_lengths):
        assert text_lengths.dim() == 1, text_lengths.shape
        assert speech.shape[0] == speech_lengths.shape[0] == text.shape[0] == text_lengths.shape[0]
        batch_size = speech.shape[0]
        tl = _lens(text_lengths)
        text = text[:, : max(tl)].contiguous()
        encoder_out, encoder_out_lens = self.encode(speech, speech_lengths)
        loss_att = acc_att = loss_ctc = None
        if self.ctc_weight != 1.0:
            ys_in_pad, ys_out_pad, _ = ops.add_sos_eos(text, self.sos, self.eos, self.ignore_id)
            decoder_out, _ = self.decoder(encoder_out, encoder_out_lens, ys_in_pad, [v + 1 for v in tl])
            loss_att = self.criterion_att(decoder_out, ys_out_pad, n_tokens=sum(tl) + len(tl))
            acc_att = M.th_accuracy(self.criterion_att.correct_rows, ys_out_pad, self.ignore_id)
        if self.ctc_weight != 0.0:
            loss_ctc = self.ctc(encoder_out, encoder_out_lens.to(torch.int32), text, tl)
        if self.ctc_weight == 0.0:
            loss = loss_att
        elif self.ctc_weight == 1.0:
            loss = loss_ctc
        else:
            loss = F_.WeightedSumFn.apply(loss_ctc, loss_att, self.ctc_weight)
        stats = dict(loss=loss.detach(), loss_att=loss_att.detach() if loss_att is not None else None,
                     loss_ctc=loss_ctc.detach() if loss_ctc is not None else None, acc=acc_att, cer=None, wer=None,
                     cer_ctc=None)
        # force_gatherable (espnet2/torch_utils/device_funcs.py): scalars -> (1,) tensors on the loss device
        dev = loss.device
        stats = {k: (v.detach().reshape(1) if isinstance(v, torch.Tensor) else v) for k, v in stats.items()}
        weight = torch.tensor([batch_size], dtype=torch.int64, device=dev)
        return loss.reshape(1), stats, weight


def register_choices(asr_task_module):
    """Add the HIP classes to the reference's ClassChoices registries (espnet2/tasks/asr.py:53-101) so
    `--frontend default_mi355x --specaug specaug_mi355x --normalize global_mvn_mi355x --encoder conformer_mi355x
    --decoder transformer_mi355x` selects them."""
    asr_task_module.encoder_choices.classes["conformer_mi355x"] = ConformerEncoder
    asr_task_module.encoder_choices.classes["transformer_mi355x"] = TransformerEncoder
    asr_task_module.decoder_choices.classes["transformer_mi355x"] = TransformerDecoder
    from .rnn import RNNDecoder, RNNEncoder, VGGRNNEncoder
    asr_task_module.decoder_choices.classes["rnn_mi355x"] = RNNDecoder
    asr_task_module.encoder_choices.classes["rnn_mi355x"] = RNNEncoder
    asr_task_module.encoder_choices.classes["vgg_rnn_mi355x"] = VGGRNNEncoder
    from .frontend import DefaultFrontend
    from .layers import GlobalMVN, SpecAug, UtteranceMVN
    asr_task_module.frontend_choices.classes["default_mi355x"] = DefaultFrontend
    asr_task_module.specaug_choices.classes["specaug_mi355x"] = SpecAug
    asr_task_module.normalize_choices.classes["global_mvn_mi355x"] = GlobalMVN
    asr_task_module.normalize_choices.classes["utterance_mvn_mi355x"] = UtteranceMVN
