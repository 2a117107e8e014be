"""espnet1 RNN model surface: VGG-BLSTM(P) encoder + location-aware attention LSTM decoder + CTC on the
espnet_amd HIP kernels.

Plug-in slot: ``--model-module espnet_amd.nets.e2e_asr:E2E``
(reference: espnet/nets/pytorch_backend/e2e_asr.py:57-338; BASELINE config 4).
"""
import math

import numpy as np
import torch

from .. import functional as F_
from .. import ops
from .asr_interface import ASRInterface
from .modules import CTC
from .rnn.attentions import att_for
from .rnn.decoders import decoder_for
from .rnn.encoders import encoder_for

CTC_LOSS_THRESHOLD = 10000  # reference: e2e_asr.py:43


def get_subsample(train_args, mode, arch):
    """reference: nets_utils.py:390-468 (asr / rnn and rnn-t arches; transformer -> [1])"""
    if arch == "transformer":
        return np.array([1])
    if mode == "asr" and arch in ("rnn", "rnn-t"):
        subsample = np.ones(train_args.elayers + 1, dtype=np.int64)
        if train_args.etype.endswith("p") and not train_args.etype.startswith("vgg"):
            ss = train_args.subsample.split("_")
            for j in range(min(train_args.elayers + 1, len(ss))):
                subsample[j] = int(ss[j])
        return subsample
    raise ValueError("Invalid options: mode={}, arch={}".format(mode, arch))


def lecun_normal_init_parameters(module):
    """reference: espnet/nets/pytorch_backend/initialization.py:14-34"""
    for p in module.parameters():
        data = p.data
        if data.dim() == 1:
            data.zero_()
        elif data.dim() == 2:
            data.normal_(0, 1.0 / math.sqrt(data.size(1)))
        elif data.dim() in (3, 4):
            n = data.size(1)
            for k in data.size()[2:]:
                n *= k
            data.normal_(0, 1.0 / math.sqrt(n))
        else:
            raise NotImplementedError


def set_forget_bias_to_one(bias):
    """reference: initialization.py:49-54"""
    n = bias.size(0)
    bias.data[n // 4: n // 2].fill_(1.0)


class E2E(ASRInterface, torch.nn.Module):
    """reference: e2e_asr.py:57-338 (single encoder, no frontend)"""

    def __init__(self, idim, odim, args):
        torch.nn.Module.__init__(self)
        self.mtlalpha = args.mtlalpha
        assert 0.0 <= self.mtlalpha <= 1.0, "mtlalpha should be [0.0, 1.0]"
        self.etype = args.etype
        self.verbose = args.verbose
        args.char_list = getattr(args, "char_list", None)
        self.char_list = args.char_list
        self.outdir = args.outdir
        self.space = args.sym_space
        self.blank = args.sym_blank
        self.sos = odim - 1
        self.eos = odim - 1
        self.subsample = get_subsample(args, mode="asr", arch="rnn")
        if getattr(args, "lsm_type", ""):
            raise NotImplementedError("unigram label smoothing needs the training json (out of the hot-path scope)")
        if getattr(args, "use_frontend", False):
            raise NotImplementedError("speech-enhancement frontend is out of the hot-path scope")
        self.frontend = None
        self.enc = encoder_for(args, idim, self.subsample)
        self.ctc = CTC(odim, args.eprojs, args.dropout_rate, ctc_type=args.ctc_type)
        self.att = att_for(args)
        self.dec = decoder_for(args, odim, self.sos, self.eos, self.att, None)
        self.init_like_chainer()
        self.report_cer = False
        self.report_wer = False
        self.rnnlm = None
        self.logzero = -10000000000.0
        self.loss = None
        self.acc = None

    def init_like_chainer(self):
        """reference: e2e_asr.py:187-203"""
        lecun_normal_init_parameters(self)
        self.dec.embed.weight.data.normal_(0, 1)
        for i in range(len(self.dec.decoder)):
            set_forget_bias_to_one(self.dec.decoder[i].bias_ih)

    def forward(self, xs_pad, ilens, ys_pad):
        """xs_pad (B,Tmax,idim), ilens (B), ys_pad (B,Lmax) -> loss (e2e_asr.py:205-338)"""
        if xs_pad.is_cuda:
            if self.training and torch.is_grad_enabled():
                ops.zero_arena_begin(xs_pad.device)      # the decoder loop's zero-filled buffers: one fill per step
            else:
                ops.zero_arena_off()
        hs_pad, hlens, _ = self.enc(xs_pad, ilens)
        self.hs_pad, self.hlens = hs_pad, hlens
        self.loss_ctc = None if self.mtlalpha == 0 else self.ctc(hs_pad, hlens, ys_pad)
        if self.mtlalpha == 1:
            self.loss_att, acc = None, None
        else:
            self.loss_att, acc, _ = self.dec(hs_pad, hlens, ys_pad)
        self.acc = acc
        alpha = self.mtlalpha
        if alpha == 0:
            self.loss = self.loss_att
        elif alpha == 1:
            self.loss = self.loss_ctc
        else:
            self.loss = F_.WeightedSumFn.apply(self.loss_ctc, self.loss_att, alpha)
        return self.loss

    def scorers(self):
        from .ctc_prefix_score import CTCPrefixScorer
        return dict(decoder=self.dec, ctc=CTCPrefixScorer(self.ctc, self.eos))

    def encode(self, x):
        """x ndarray (T, D) -> encoder states (T', eprojs) (e2e_asr.py:344-369)"""
        self.eval()
        ops.zero_arena_off()                    # recognition: no slices of a training step's zero arena
        p = next(self.parameters())
        h = torch.as_tensor(x, device=p.device, dtype=p.dtype).unsqueeze(0)
        with torch.no_grad():
            hs, _, _ = self.enc(h, [x.shape[0]])
        return hs.squeeze(0)

    @ops.inference_call
    def recognize_batch(self, xs, recog_args, char_list=None, rnnlm=None, ctc_scoring_num=None):
        """xs list of ndarrays (T_b, D) -> per utterance an n-best list (e2e_asr.py:394-445 -> Decoder.recognize_beam_batch)"""
        self.eval()
        ops.zero_arena_off()
        p = next(self.parameters())
        ilens = [int(x.shape[0]) for x in xs]
        feats = [torch.as_tensor(x, device=p.device, dtype=p.dtype) for x in xs]
        xs_pad = torch.nn.utils.rnn.pad_sequence(feats, batch_first=True)
        with torch.no_grad():
            hs_pad, hlens, _ = self.enc(xs_pad, ilens)
            if recog_args.ctc_weight > 0.0:
                lpz, normalize = self.ctc.log_softmax(hs_pad), False
            else:
                lpz, normalize = None, True
            return self.dec.recognize_beam_batch(hs_pad, hlens, lpz, recog_args, char_list, rnnlm, normalize_score=normalize,
                                                 ctc_scoring_num=ctc_scoring_num)

    @ops.inference_call
    def recognize(self, x, recog_args, char_list=None, rnnlm=None):
        """x ndarray (T, D) -> n-best list of {"score", "yseq"} (e2e_asr.py:372-392)"""
        hs = self.encode(x).unsqueeze(0)
        lpz = self.ctc.log_softmax(hs)[0] if recog_args.ctc_weight > 0.0 else None
        return self.dec.recognize_beam(hs[0], lpz, recog_args, char_list, rnnlm)
