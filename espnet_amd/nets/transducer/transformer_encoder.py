"""reference: espnet/nets/pytorch_backend/transducer/transformer_encoder.py:12-91"""
import torch

from ..modules import LayerNorm
from .blocks import build_blocks


class Encoder(torch.nn.Module):
    """Custom (block-list) Transformer / Conformer encoder of the transducer model."""

    def __init__(self, idim, enc_arch, input_layer="linear", repeat_block=0, self_attn_type="selfattn",
                 positional_encoding_type="abs_pos", positionwise_layer_type="linear",
                 positionwise_activation_type="relu", conv_mod_activation_type="relu", normalize_before=True,
                 padding_idx=-1):
        super().__init__()
        self.embed, self.encoders, self.enc_out = build_blocks(
            "encoder", idim, input_layer, enc_arch, repeat_block=repeat_block, self_attn_type=self_attn_type,
            positional_encoding_type=positional_encoding_type, positionwise_layer_type=positionwise_layer_type,
            positionwise_activation_type=positionwise_activation_type,
            conv_mod_activation_type=conv_mod_activation_type, padding_idx=padding_idx)
        self.normalize_before = normalize_before
        if self.normalize_before:
            self.after_norm = LayerNorm(self.enc_out)

    def forward(self, xs, masks):
        xs, masks = self.embed(xs, masks)
        xs, masks = self.encoders(xs, masks)
        if isinstance(xs, tuple):
            xs = xs[0]
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, masks
