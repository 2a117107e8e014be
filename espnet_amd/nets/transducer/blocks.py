"""Block builders of the transducer's custom encoder.  reference:
espnet/nets/pytorch_backend/transducer/blocks.py:39-552 (check_and_prepare, get_pos_enc_and_att_class,
build_input_layer, build_transformer_block, build_conformer_block, build_blocks).
Block types on the HIP path: "transformer" and "conformer" encoder blocks behind a conv2d input layer
(the configuration BASELINE config 5 names) and "transformer" decoder blocks behind an embed input layer
(transformer-transducer); tdnn / causal-conv1d / vgg2l inputs raise."""
from collections import Counter

from ..modules import (ConformerEncoderLayer, Conv2dSubsampling, ConvolutionModule, MultiHeadedAttention,
                       MultiSequential, PositionalEncoding, PositionwiseFeedForward, RelPositionalEncoding,
                       RelPositionMultiHeadedAttention, TransformerEncoderLayer, get_activation)


def _most_common_max(blocks_arch, key):
    c = sorted(Counter(b[key] for b in blocks_arch if key in b).most_common(), key=lambda x: x[0], reverse=True)
    return c[0][0] if c else 0.0


def check_and_prepare(net_part, blocks_arch, input_layer):
    """reference: blocks.py:39-222"""
    if blocks_arch[0]["type"] in ("tdnn", "causal-conv1d"):
        raise NotImplementedError("tdnn / causal-conv1d blocks have no HIP kernels")
    input_layer_odim = blocks_arch[0]["d_hidden"]
    input_dropout_rate = _most_common_max(blocks_arch, "dropout-rate")
    input_pos_dropout_rate = _most_common_max(blocks_arch, "pos-dropout-rate")
    has_transformer = has_conformer = False
    cmp_io = []
    for i, b in enumerate(blocks_arch):
        if "type" not in b:
            raise ValueError("type is not defined in the " + str(i + 1) + "th block.")
        t = b["type"]
        if t == "transformer":
            if not {"d_hidden", "d_ff", "heads"}.issubset(b):
                raise ValueError("Block %d in %s: Transformer block format is: {'type: transformer', "
                                 "'d_hidden': int, 'd_ff': int, 'heads': int, [...]}" % (i + 1, net_part))
            has_transformer = True
        elif t == "conformer":
            if not {"d_hidden", "d_ff", "heads", "macaron_style", "use_conv_mod"}.issubset(b):
                raise ValueError("Block %d in %s: Conformer block format is {'type: conformer', 'd_hidden': int, "
                                 "'d_ff': int, 'heads': int, 'macaron_style': bool, 'use_conv_mod': bool, [...]}"
                                 % (i + 1, net_part))
            if b["use_conv_mod"] is True and "conv_mod_kernel" not in b:
                raise ValueError("Block %d: 'use_conv_mod' is True but 'use_conv_kernel' is not specified" % (i + 1))
            if i == 0 and input_layer == "conv2d":
                input_layer = "conformer-conv2d"
            has_conformer = True
        else:
            raise NotImplementedError("Block %d in %s: type %r has no HIP kernels" % (i + 1, net_part, t))
        cmp_io.append((b["d_hidden"], b["d_hidden"]))
    if has_transformer and has_conformer:
        raise NotImplementedError(net_part + ": transformer and conformer blocks can't be defined in the same net part.")
    for i in range(1, len(cmp_io)):
        if cmp_io[i - 1][1] != cmp_io[i][0]:
            raise ValueError("Output/Input mismatch between blocks %d and %d in %s" % (i, i + 1, net_part))
    return input_layer, input_layer_odim, input_dropout_rate, input_pos_dropout_rate, blocks_arch[-1]["d_hidden"]


def get_pos_enc_and_att_class(net_part, pos_enc_type, self_attn_type):
    """reference: blocks.py:225-259"""
    if pos_enc_type == "abs_pos":
        pos_enc_class = PositionalEncoding
    elif pos_enc_type == "rel_pos":
        if net_part == "encoder" and self_attn_type != "rel_self_attn":
            raise ValueError("'rel_pos' is only compatible with 'rel_self_attn'")
        pos_enc_class = RelPositionalEncoding
    else:
        raise NotImplementedError("pos_enc_type should be either 'abs_pos' or 'rel_pos' on the HIP path")
    attn = RelPositionMultiHeadedAttention if self_attn_type == "rel_self_attn" else MultiHeadedAttention
    return pos_enc_class, attn


def build_input_layer(input_layer, idim, odim, pos_enc_class, dropout_rate_embed, dropout_rate, pos_dropout_rate,
                      padding_idx):
    """reference: blocks.py:262-321"""
    if input_layer == "conv2d":
        return Conv2dSubsampling(idim, odim, dropout_rate)
    if input_layer == "conformer-conv2d":
        return Conv2dSubsampling(idim, odim, dropout_rate, pos_enc_class(odim, pos_dropout_rate))
    if input_layer == "embed":      # decoder side (blocks.py:290-294)
        from ..modules import _EmbedInput
        return _EmbedInput(idim, odim, padding_idx, pos_enc_class(odim, pos_dropout_rate))
    raise NotImplementedError("input layer %r: conv2d (encoder) and embed (decoder) are on the HIP path" % (input_layer,))


def _rates(block_arch):
    return (block_arch.get("dropout-rate", 0.0), block_arch.get("pos-dropout-rate", 0.0),
            block_arch.get("att-dropout-rate", 0.0))


def build_transformer_block(net_part, block_arch, pw_layer_type, pw_activation_type):
    """reference: blocks.py:324-364"""
    d_hidden, d_ff, heads = block_arch["d_hidden"], block_arch["d_ff"], block_arch["heads"]
    dropout_rate, pos_dropout_rate, att_dropout_rate = _rates(block_arch)
    if pw_layer_type != "linear":
        raise NotImplementedError("Transformer block only supports linear yet.")
    return lambda: TransformerEncoderLayer(
        d_hidden, MultiHeadedAttention(heads, d_hidden, att_dropout_rate),
        PositionwiseFeedForward(d_hidden, d_ff, pos_dropout_rate, get_activation(pw_activation_type)), dropout_rate)


def build_conformer_block(block_arch, self_attn_class, pos_enc_class, pw_layer_type, pw_activation_type,
                          conv_mod_activation_type):
    """reference: blocks.py:367-422"""
    d_hidden, d_ff, heads = block_arch["d_hidden"], block_arch["d_ff"], block_arch["heads"]
    macaron_style, use_conv_mod = block_arch["macaron_style"], block_arch["use_conv_mod"]
    dropout_rate, pos_dropout_rate, att_dropout_rate = _rates(block_arch)
    if pw_layer_type != "linear":
        raise NotImplementedError("Conformer block only supports linear yet.")

    def pw():
        return PositionwiseFeedForward(d_hidden, d_ff, pos_dropout_rate, get_activation(pw_activation_type))

    return lambda: ConformerEncoderLayer(
        d_hidden, self_attn_class(heads, d_hidden, att_dropout_rate), pw(), pw() if macaron_style else None,
        ConvolutionModule(d_hidden, block_arch["conv_mod_kernel"], get_activation(conv_mod_activation_type))
        if use_conv_mod else None, dropout_rate)


def build_blocks(net_part, idim, input_layer, blocks_arch, repeat_block=0, self_attn_type="self_attn",
                 positional_encoding_type="abs_pos", positionwise_layer_type="linear",
                 positionwise_activation_type="relu", conv_mod_activation_type="relu", dropout_rate_embed=0.0,
                 padding_idx=-1):
    """reference: blocks.py:463-552 -> (input layer, MultiSequential of blocks, output dim)"""
    input_layer, input_layer_odim, input_dropout_rate, input_pos_dropout_rate, out_dim = \
        check_and_prepare(net_part, blocks_arch, input_layer)
    pos_enc_class, self_attn_class = get_pos_enc_and_att_class(net_part, positional_encoding_type, self_attn_type)
    in_layer = build_input_layer(input_layer, idim, input_layer_odim, pos_enc_class, dropout_rate_embed,
                                 input_dropout_rate, input_pos_dropout_rate, padding_idx)
    fn_modules = []
    for b in blocks_arch:
        if b["type"] == "transformer":
            fn_modules.append(build_transformer_block(net_part, b, positionwise_layer_type,
                                                      positionwise_activation_type))
        else:
            fn_modules.append(build_conformer_block(b, self_attn_class, pos_enc_class, positionwise_layer_type,
                                                    positionwise_activation_type, conv_mod_activation_type))
    if repeat_block > 1:
        fn_modules = fn_modules * repeat_block
    return in_layer, MultiSequential([fn() for fn in fn_modules]), out_dim
