"""espnet1 model surface: hybrid CTC/attention Transformer ASR on the HIP kernels.

Plug-in slot: ``--model-module espnet_amd.nets.e2e_asr_transformer:E2E``
(reference: espnet/nets/pytorch_backend/e2e_asr_transformer.py:38-525; resolved by
espnet/utils/dynamic_import.py:4-23 at espnet/asr/pytorch_backend/asr.py:437-442).
"""
import argparse
import math
from itertools import groupby

import numpy as np
import torch

from .. import functional as F_
from .. import ops
from .asr_interface import ASRInterface
from .modules import (CTC, Decoder, LabelSmoothingLoss, TransformerEncoder, make_non_pad_mask,
                      embed_output_lengths, grad_cut, subsampled_lengths, subsampled_stride, subsequent_mask, th_accuracy)

CTC_LOSS_THRESHOLD = 10000  # reference: e2e_asr.py:43


def strtobool(v):
    v = str(v).lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if v in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError("invalid truth value %r" % (v,))


def add_arguments_transformer_common(group):
    """Same option names / defaults as transformer/argument.py:10-160 (only options on the path)."""
    group.add_argument("--transformer-init", type=str, default="pytorch")
    group.add_argument("--transformer-input-layer", type=str, default="conv2d",
                       choices=["conv2d", "linear", "embed"])
    group.add_argument("--transformer-attn-dropout-rate", default=None, type=float)
    group.add_argument("--transformer-lr", default=10.0, type=float)
    group.add_argument("--transformer-warmup-steps", default=25000, type=int)
    group.add_argument("--transformer-length-normalized-loss", default=True, type=strtobool)
    group.add_argument("--transformer-encoder-selfattn-layer-type", type=str, default="selfattn")
    group.add_argument("--transformer-decoder-selfattn-layer-type", type=str, default="selfattn")
    group.add_argument("--wshare", default=4, type=int)
    group.add_argument("--ldconv-encoder-kernel-length", default="21_23_25_27_29_31_33_35_37_39_41_43", type=str)
    group.add_argument("--ldconv-decoder-kernel-length", default="11_13_15_17_19_21", type=str)
    group.add_argument("--ldconv-usebias", type=strtobool, default=False)
    group.add_argument("--dropout-rate", default=0.0, type=float)
    group.add_argument("--elayers", default=4, type=int)
    group.add_argument("--eunits", "-u", default=300, type=int)
    group.add_argument("--adim", default=320, type=int)
    group.add_argument("--aheads", default=4, type=int)
    group.add_argument("--dlayers", default=1, type=int)
    group.add_argument("--dunits", default=320, type=int)
    return group


_DEFAULTS = dict(mtlalpha=0.3, lsm_weight=0.0, ctc_type="warpctc", report_cer=False, report_wer=False,
                 char_list=None, sym_space="<space>", sym_blank="<blank>")


def fill_missing_args(args, add_arguments):
    """reference: espnet/utils/fill_missing_args.py (defaults for options absent from `args`)."""
    parser = argparse.ArgumentParser()
    add_arguments(parser)
    defaults, _ = parser.parse_known_args([])
    merged = dict(_DEFAULTS)
    merged.update(vars(defaults))
    merged.update(vars(args))
    return argparse.Namespace(**merged)


class Reporter:
    """Stand-in for the chainer reporter (e2e_asr.py:46-58): keeps the last values, no host sync."""

    def __init__(self):
        self.last = {}

    def report(self, loss_ctc, loss_att, acc, cer_ctc, cer, wer, mtl_loss):
        self.last = dict(loss_ctc=loss_ctc, loss_att=loss_att, acc=acc, cer_ctc=cer_ctc, cer=cer, wer=wer,
                         loss=mtl_loss)


class E2E(ASRInterface, torch.nn.Module):
    """E2E module (reference: e2e_asr_transformer.py:38-525)."""

    @staticmethod
    def add_arguments(parser):
        group = parser.add_argument_group("transformer model setting")
        add_arguments_transformer_common(group)
        return parser

    def get_total_subsampling_factor(self):
        return 4  # conv2d input layer (e2e_asr_transformer.py:66-68)

    def _build_encoder(self, idim, args):
        if args.transformer_encoder_selfattn_layer_type != "selfattn":
            raise NotImplementedError("encoder selfattention_layer_type must be selfattn for the Transformer E2E")
        return TransformerEncoder(
            idim=idim, attention_dim=args.adim, attention_heads=args.aheads, linear_units=args.eunits,
            num_blocks=args.elayers, input_layer=args.transformer_input_layer, dropout_rate=args.dropout_rate,
            positional_dropout_rate=args.dropout_rate, attention_dropout_rate=args.transformer_attn_dropout_rate)

    def __init__(self, idim, odim, args, ignore_id=-1):
        torch.nn.Module.__init__(self)
        args = fill_missing_args(args, self.add_arguments)
        if args.transformer_attn_dropout_rate is None:
            args.transformer_attn_dropout_rate = args.dropout_rate
        self.encoder = self._build_encoder(idim, args)
        if args.mtlalpha < 1:
            self.decoder = Decoder(
                odim=odim, selfattention_layer_type=args.transformer_decoder_selfattn_layer_type,
                attention_dim=args.adim, attention_heads=args.aheads, linear_units=args.dunits,
                num_blocks=args.dlayers, dropout_rate=args.dropout_rate, positional_dropout_rate=args.dropout_rate,
                self_attention_dropout_rate=args.transformer_attn_dropout_rate,
                src_attention_dropout_rate=args.transformer_attn_dropout_rate)
            self.criterion = LabelSmoothingLoss(odim, ignore_id, args.lsm_weight,
                                                bool(args.transformer_length_normalized_loss))
        else:
            self.decoder = None
            self.criterion = None
        self.blank = 0
        self.sos = odim - 1
        self.eos = odim - 1
        self.odim = odim
        self.ignore_id = ignore_id
        self.subsample = np.array([1])  # get_subsample(arch="transformer") (nets_utils.py:390-468)
        self.reporter = Reporter()
        self.adim = args.adim
        self.mtlalpha = args.mtlalpha
        if args.mtlalpha > 0.0:
            self.ctc = CTC(odim, args.adim, args.dropout_rate, ctc_type=args.ctc_type, reduce=True)
        else:
            self.ctc = None
        self.error_calculator = None   # host-side edit distance: out of scope
        self.rnnlm = None
        self.sync_report = True        # float(loss) x3 like the reference; False defers the D2H copies
        self.reset_parameters(args)

    def reset_parameters(self, args):
        """reference: transformer/initializer.py:12-45"""
        init_type = args.transformer_init
        if init_type == "pytorch":
            return
        fn = {"xavier_uniform": torch.nn.init.xavier_uniform_, "xavier_normal": torch.nn.init.xavier_normal_,
              "kaiming_uniform": lambda p: torch.nn.init.kaiming_uniform_(p, nonlinearity="relu"),
              "kaiming_normal": lambda p: torch.nn.init.kaiming_normal_(p, nonlinearity="relu")}
        if init_type not in fn:
            raise ValueError("Unknown initialization: " + init_type)
        for p in self.parameters():
            if p.dim() > 1:
                fn[init_type](p.data)
        for p in self.parameters():
            if p.dim() == 1:
                p.data.zero_()
        for m in self.modules():
            if isinstance(m, (torch.nn.Embedding, torch.nn.LayerNorm)):
                m.reset_parameters()

    # ---- training forward ---------------------------------------------------------------------
    def prepare(self, xs_pad, ilens, ys_pad, pad_to=None):
        """Host-side part of forward (lengths, masks, <sos>/<eos>): everything that needs Python lists
        or H2D copies.  The returned dict feeds forward_core(), which only launches kernels and can be
        captured into a hipGraph.  reference: e2e_asr_transformer.py:173-183,202.
        pad_to = (T, L): instead of cropping to the longest utterance, pad frames (zeros) and labels (ignore_id) up to
        these sizes - the batch then has the shapes of its bucket (train.BucketedGraphStep)."""
        il = [int(v) for v in (ilens.tolist() if isinstance(ilens, torch.Tensor) else ilens)]
        tmax = max(il)
        dev = next(self.parameters()).device
        hl_true = tbound = None
        if pad_to is not None:
            Tb, Lb = pad_to
            assert Tb >= tmax
            # what the reference computes on this batch cropped to its own longest utterance: the valid encoder frames of
            # every utterance and the length T' of the encoder's time axis (= those of the longest utterance)
            hl_true = embed_output_lengths(self.encoder.embed, il, tmax)
            tbound = embed_output_lengths(self.encoder.embed, [tmax], tmax)[0]
            xp = xs_pad.new_zeros(xs_pad.shape[0], Tb, xs_pad.shape[2])
            xp[:, :tmax] = xs_pad[:, :tmax]
            yp = ys_pad.new_full((ys_pad.shape[0], Lb), self.ignore_id)
            n = min(Lb, ys_pad.shape[1])
            yp[:, :n] = ys_pad[:, :n]
            xs_pad, ys_pad, tmax = xp, yp, Tb
        # host-built tensors go to the device through pinned staging copies (ops.h2d_async): the host never waits for the
        # stream here, so the next batch is prepared while the previous step still runs
        xs_pad = ops.h2d_async(xs_pad[:, :tmax], dev).contiguous()
        ys_pad = ops.h2d_async(ys_pad, dev).contiguous()
        if hl_true is None:
            mask_len = il
        else:
            # the input layer subsamples the mask by plain slicing (encoder frame t' <- input frame t' * stride): a padded batch
            # gets the input mask whose slices are exactly the reference's encoder mask (t' < hl_true[b])
            stride = max(1, subsampled_stride(self.encoder.embed))
            mask_len = [(h - 1) * stride + 1 if h > 0 else 0 for h in hl_true]
        src_mask = ops.h2d_async(make_non_pad_mask(mask_len, tmax).unsqueeze(-2).to(torch.uint8), dev)     # (B,1,T)
        batch = dict(xs_pad=xs_pad, ys_pad=ys_pad, src_mask=src_mask, B=xs_pad.size(0))
        if tbound is not None:
            batch["tbound"] = ops.h2d_async(torch.tensor([tbound], dtype=torch.int32), dev)
        if self.decoder is not None:
            ys_in_pad, ys_out_pad, _ = ops.add_sos_eos(ys_pad, self.sos, self.eos, self.ignore_id)
            U = ys_in_pad.size(1)
            # ys_in is padded with <eos>, never ignore_id, so target_mask() is the causal mask (mask.py:41-51)
            ys_mask = subsequent_mask(U).unsqueeze(0).expand(xs_pad.size(0), U, U).to(torch.uint8).contiguous()
            batch.update(ys_in_pad=ys_in_pad, ys_out_pad=ys_out_pad, ys_mask=ops.h2d_async(ys_mask, dev),
                         n_valid=(ys_out_pad != self.ignore_id).sum())
        if self.mtlalpha > 0.0:
            hl = hl_true if hl_true is not None else embed_output_lengths(self.encoder.embed, il, tmax)
            batch["hs_len"] = ops.h2d_async(torch.tensor(hl, dtype=torch.int32), dev)
        return batch

    def forward_core(self, batch):
        """Kernel-only part of forward (reference: e2e_asr_transformer.py:175-232)."""
        xs_pad = batch["xs_pad"]
        # a batch padded by a shape bucket carries the length of its own encoder time axis: rel_shift, the depthwise
        # convolution's zero padding and the BatchNorm statistics follow it (ops.set_time_bound); None otherwise
        ops.set_time_bound(batch.get("tbound"))
        try:
            hs_pad, hs_mask = self.encoder(xs_pad, batch["src_mask"])
        finally:
            ops.set_time_bound(None)       # the blocks keep the bound they read for their backward
        hs_pad = grad_cut(hs_pad)        # no-op unless a phased backward is being set up (modules.GradCuts)
        self.hs_pad = hs_pad
        if hs_mask is not None and not hs_mask.is_contiguous():
            hs_mask = hs_mask.contiguous()
        loss_att = loss_ctc = None
        self._acc_t = None
        if self.decoder is not None:
            pred_pad, _ = self.decoder(batch["ys_in_pad"], batch["ys_mask"], hs_pad, hs_mask)
            self.pred_pad = pred_pad
            loss_att = self.criterion(pred_pad, batch["ys_out_pad"])
            self._acc_t = ops.reduce_sum(self.criterion.correct_rows) / batch["n_valid"]
        if self.mtlalpha > 0.0:
            loss_ctc = self.ctc(hs_pad.view(batch["B"], -1, self.adim), batch["hs_len"], batch["ys_pad"])
        alpha = self.mtlalpha
        if alpha == 0:
            self.loss = loss_att
        elif alpha == 1:
            self.loss = loss_ctc
        else:
            self.loss = F_.WeightedSumFn.apply(loss_ctc, loss_att, alpha)
        self._loss_ctc_t, self._loss_att_t = loss_ctc, loss_att
        return self.loss

    def forward(self, xs_pad, ilens, ys_pad):
        """reference: e2e_asr_transformer.py:159-241.  Returns the 0-dim loss tensor."""
        self.loss = self.forward_core(self.prepare(xs_pad, ilens, ys_pad))
        self.acc = None
        if self.sync_report:
            self._report()
        return self.loss

    def _report(self):
        """Host copies of the scalars (reference does float(loss) x3 inside forward)."""
        lc = float(self._loss_ctc_t.detach()) if self._loss_ctc_t is not None else None
        la = float(self._loss_att_t.detach()) if self._loss_att_t is not None else None
        self.acc = float(self._acc_t) if self.decoder is not None else None
        loss_data = float(self.loss.detach())
        if loss_data < CTC_LOSS_THRESHOLD and not math.isnan(loss_data):
            self.reporter.report(lc, la, self.acc, None, None, None, loss_data)
        return loss_data

    # ---- inference ------------------------------------------------------------------------------
    def scorers(self):
        from .ctc_prefix_score import CTCPrefixScorer
        return dict(decoder=self.decoder, ctc=CTCPrefixScorer(self.ctc, self.eos))

    def encode(self, x):
        """x: (T, idim) ndarray / tensor -> (T', adim) tensor (e2e_asr_transformer.py:247-257)"""
        self.eval()
        dev = next(self.parameters()).device
        x = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x, dtype=torch.float32)
        x = x.to(dev).unsqueeze(0)
        with torch.no_grad():
            enc_output, _ = self.encoder(x, None)
        return enc_output.squeeze(0)

    @ops.inference_call
    def recognize(self, x, recog_args, char_list=None, rnnlm=None, use_jit=False):
        """reference: e2e_asr_transformer.py:259-477 (greedy CTC when ctc_weight == 1, else joint
        CTC/attention beam search through the scorer interface)."""
        enc_output = self.encode(x).unsqueeze(0)
        if self.mtlalpha == 1.0:
            recog_args.ctc_weight = 1.0
        if self.mtlalpha > 0 and recog_args.ctc_weight == 1.0:
            with torch.no_grad():
                ids = self.ctc.argmax(enc_output).to(torch.int32)
                hyp, n = ops.ctc_collapse(ids.contiguous(), None, self.blank)
            hyp = hyp[0, : int(n[0])].tolist()
            if recog_args.beam_size > 1:
                raise NotImplementedError("Pure CTC beam search is not implemented.")
            return [{"score": 0.0, "yseq": [self.sos] + hyp}]
        from .beam_search import recognize_beam
        return recognize_beam(self, enc_output.squeeze(0), recog_args, char_list, rnnlm)

    def greedy_ctc_batch(self, xs_pad, ilens):
        """Batched greedy CTC (embarrassingly parallel per utterance): token ids [B, T'] padded -1."""
        il = [int(v) for v in (ilens.tolist() if isinstance(ilens, torch.Tensor) else ilens)]
        tmax = max(il)
        with torch.no_grad():
            hs, _ = self.encoder(xs_pad[:, :tmax], make_non_pad_mask(il).unsqueeze(-2))
            ids = self.ctc.argmax(hs).to(torch.int32).contiguous()
            hl = torch.tensor(embed_output_lengths(self.encoder.embed, il, tmax), dtype=torch.int32).to(ids.device)
            return ops.ctc_collapse(ids, hl, self.blank)

    def calculate_all_attentions(self, xs_pad, ilens, ys_pad):
        """reference: e2e_asr_transformer.py:479-503.  {module name: (B, H, T1, T2) float ndarray} for every attention
        module (encoder self-attention, decoder self- and source-attention) from one eval-mode forward pass."""
        from .modules import MultiHeadedAttention
        self.eval()
        F_.ATTN_TAP = []
        try:
            with torch.no_grad():
                self.forward(xs_pad, ilens, ys_pad)
        finally:
            F_.ATTN_TAP = None
        ret = dict()
        for name, m in self.named_modules():
            if isinstance(m, MultiHeadedAttention) and m.attn is not None:
                ret[name] = m.attn.cpu().numpy()
        self.train()       # the reference leaves the model in training mode (e2e_asr_transformer.py:502)
        return ret

    def calculate_all_ctc_probs(self, xs_pad, ilens, ys_pad):
        """reference: e2e_asr_transformer.py:503-525"""
        if self.mtlalpha == 0:
            return None
        self.eval()
        with torch.no_grad():
            self.forward(xs_pad, ilens, ys_pad)
            ret = self.ctc.softmax(self.hs_pad).cpu().numpy()
        self.train()
        return ret
