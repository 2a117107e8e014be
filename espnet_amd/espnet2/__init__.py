"""espnet2 model surface on the HIP kernels (reference: espnet2/asr/*)."""
from .asr import CTC, ConformerEncoder, ESPnetASRModel, TransformerDecoder, TransformerEncoder, register_choices  # noqa: F401
from .layers import GlobalMVN, MaskAlongAxis, SpecAug, TimeWarp, UtteranceMVN  # noqa: F401,E402
from .lm import SequentialRNNLM, TransformerLM, register_lm_choices  # noqa: F401,E402
from .asr_inference import Speech2Text  # noqa: F401,E402
from .frontend import DefaultFrontend, LogMel, Stft  # noqa: F401,E402
from .rnn import RNNDecoder, RNNEncoder, VGGRNNEncoder  # noqa: F401,E402
