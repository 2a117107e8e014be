"""Transformer-transducer prediction network + joint network on the HIP kernels.

reference: espnet/nets/pytorch_backend/transducer/transformer_decoder.py:16-283 (DecoderTT),
transformer_decoder_layer.py:9-75 (self-attention + feed-forward layer with an incremental cache - the same arithmetic
as the encoder layer, which is what runs here), transducer/utils.py:206-285 (pad_sequence, check_state,
pad_batch_state).  The reference's decoding conventions are kept as they are: the training mask hides blank keys
(target_mask), the scoring masks are causal only, and batched scoring left-pads prefixes / cached states with blank / 0.
"""
import torch

from .. import modules as M
from .blocks import build_blocks
from .joint_network import JointNetwork


def pad_sequence(seqlist, pad_token):
    """left-pad token id sequences (utils.py:206-221)"""
    maxlen = max(len(x) for x in seqlist)
    return [([pad_token] * (maxlen - len(x))) + x for x in seqlist]


def check_state(state, max_len, pad_token):
    """left-pad / trim the L cached layer outputs (1, len, D) to max_len (utils.py:224-262)"""
    if state is None or max_len < 1 or state[0].size(1) == max_len:
        return state
    curr_len = state[0].size(1)
    if curr_len > max_len:
        trim_val = int(curr_len - max_len)
        return [s[:, trim_val:, :] for s in state]
    final = [s.new_full((1, max_len, s.size(2)), float(pad_token)) for s in state]
    for i, s in enumerate(state):
        final[i][:, (max_len - s.size(1)):max_len, :] = s
    return final


def pad_batch_state(state, pred_length, pad_token):
    """list of (len_i, D) cached outputs of one layer -> (B, pred_length - 1, D), left-padded (utils.py:265-285)"""
    batch = len(state)
    maxlen = max(s.size(0) for s in state)
    final = state[0].new_full((batch, maxlen, state[0].size(1)), float(pad_token))
    for i, s in enumerate(state):
        final[i, (maxlen - s.size(0)):maxlen, :] = s
    trim_val = final[0].size(0) - (pred_length - 1)
    return final[:, trim_val:, :]


class DecoderTT(torch.nn.Module):
    def __init__(self, odim, edim, jdim, dec_arch, input_layer="embed", repeat_block=0, joint_activation_type="tanh",
                 positional_encoding_type="abs_pos", positionwise_layer_type="linear",
                 positionwise_activation_type="relu", dropout_rate_embed=0.0, blank=0):
        super().__init__()
        self.embed, self.decoders, ddim = build_blocks(
            "decoder", odim, input_layer, dec_arch, repeat_block=repeat_block,
            positional_encoding_type=positional_encoding_type, positionwise_layer_type=positionwise_layer_type,
            positionwise_activation_type=positionwise_activation_type, dropout_rate_embed=dropout_rate_embed,
            padding_idx=blank)
        self.after_norm = M.LayerNorm(ddim)
        self.joint_network = JointNetwork(odim, edim, ddim, jdim, joint_activation_type)
        self.dunits, self.odim, self.blank = ddim, odim, blank

    def init_state(self, init_tensor=None):
        return [None] * len(self.decoders)

    def forward(self, tgt, tgt_mask, memory):
        """tgt (B, U) token ids, tgt_mask (B, U, U), memory (B, T, D_enc) -> joint logits (B, T, U, odim), mask"""
        x = self.embed(tgt)
        x, tgt_mask = self.decoders(x, tgt_mask)
        x = self.after_norm(x.contiguous())
        return self.joint_network(memory.unsqueeze(2), x.unsqueeze(1)), tgt_mask

    def _step(self, tokens, state):
        """tokens (n, L) -> (outputs of the last position (n, D), new per-layer caches (n, L, D))"""
        dev = self.after_norm.weight.device
        mask = M.subsequent_mask(tokens.size(-1), device=dev).unsqueeze(0).expand(tokens.size(0), -1, -1).contiguous()
        x = self.embed(tokens)
        new_state = []
        for s, decoder in zip(state, self.decoders):
            x, mask = decoder(x, mask, cache=s)
            new_state.append(x)
        return self.after_norm(x[:, -1].contiguous()), new_state

    def score(self, hyp, cache, init_tensor=None):
        """reference: transformer_decoder.py:119-156"""
        dev = self.after_norm.weight.device
        tgt = torch.tensor(hyp.yseq, dtype=torch.long).to(dev).unsqueeze(0)
        lm_tokens = tgt[:, -1]
        str_yseq = "".join([str(x) for x in hyp.yseq])
        if str_yseq in cache:
            y, new_state = cache[str_yseq]
        else:
            state = check_state(hyp.dec_state, (tgt.size(1) - 1), self.blank)
            y, new_state = self._step(tgt, state)
            cache[str_yseq] = (y, new_state)
        return y, new_state, lm_tokens
