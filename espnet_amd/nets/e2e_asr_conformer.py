"""espnet1 model surface: Conformer hybrid CTC/attention ASR on the HIP kernels.

Plug-in slot: ``--model-module espnet_amd.nets.e2e_asr_conformer:E2E``
(reference: espnet/nets/pytorch_backend/e2e_asr_conformer.py:20-70).
"""
from .e2e_asr_transformer import E2E as E2ETransformer
from .e2e_asr_transformer import strtobool
from .modules import ConformerEncoder


def add_arguments_conformer_common(group):
    """Same option names / defaults as conformer/argument.py:10-45."""
    group.add_argument("--transformer-encoder-pos-enc-layer-type", type=str, default="abs_pos",
                       choices=["abs_pos", "scaled_abs_pos", "rel_pos"])
    group.add_argument("--transformer-encoder-activation-type", type=str, default="swish",
                       choices=["relu", "hardtanh", "selu", "swish"])
    group.add_argument("--macaron-style", default=False, type=strtobool)
    group.add_argument("--use-cnn-module", default=False, type=strtobool)
    group.add_argument("--cnn-module-kernel", default=31, type=int)
    return group


class E2E(E2ETransformer):
    """E2E module (reference: e2e_asr_conformer.py:20-70)."""

    @staticmethod
    def add_arguments(parser):
        E2ETransformer.add_arguments(parser)
        E2E.add_conformer_arguments(parser)
        return parser

    @staticmethod
    def add_conformer_arguments(parser):
        group = parser.add_argument_group("conformer model specific setting")
        add_arguments_conformer_common(group)
        return parser

    def _build_encoder(self, idim, args):
        return ConformerEncoder(
            idim=idim, attention_dim=args.adim, attention_heads=args.aheads, linear_units=args.eunits,
            num_blocks=args.elayers, input_layer=args.transformer_input_layer, dropout_rate=args.dropout_rate,
            positional_dropout_rate=args.dropout_rate, attention_dropout_rate=args.transformer_attn_dropout_rate,
            pos_enc_layer_type=args.transformer_encoder_pos_enc_layer_type,
            selfattention_layer_type=args.transformer_encoder_selfattn_layer_type,
            activation_type=args.transformer_encoder_activation_type, macaron_style=bool(args.macaron_style),
            use_cnn_module=bool(args.use_cnn_module), cnn_module_kernel=args.cnn_module_kernel)
