"""ASR interface the espnet1 trainers check for (reference: espnet/nets/asr_interface.py:9-134).

When the reference package is importable its ASRInterface is used as the base class so that
`assert isinstance(model, ASRInterface)` (espnet/asr/pytorch_backend/asr.py:442) holds; otherwise a
structurally identical stand-alone base is provided.
"""
try:  # pragma: no cover - only when the reference is installed next to us
    from espnet.nets.asr_interface import ASRInterface  # type: ignore
except Exception:  # noqa: BLE001

    class ASRInterface:
        """Minimal stand-alone equivalent (same method names and signatures)."""

        @staticmethod
        def add_arguments(parser):
            return parser

        @classmethod
        def build(cls, idim, odim, **kwargs):
            import argparse

            parser = argparse.ArgumentParser()
            cls.add_arguments(parser)
            args = parser.parse_args([])
            for k, v in kwargs.items():
                setattr(args, k, v)
            return cls(idim, odim, args)

        def forward(self, xs, ilens, ys):
            raise NotImplementedError("forward method is not implemented")

        def recognize(self, x, recog_args, char_list=None, rnnlm=None):
            raise NotImplementedError("recognize method is not implemented")

        def recognize_batch(self, x, recog_args, char_list=None, rnnlm=None):
            raise NotImplementedError("Batch decoding is not supported yet.")

        def calculate_all_attentions(self, xs, ilens, ys):
            raise NotImplementedError("calculate_all_attentions method is not implemented")

        def calculate_all_ctc_probs(self, xs, ilens, ys):
            raise NotImplementedError("calculate_all_ctc_probs method is not implemented")

        @property
        def attention_plot_class(self):
            raise NotImplementedError("plotting is host-side tooling, out of the hot-path scope")

        @property
        def ctc_plot_class(self):
            raise NotImplementedError("plotting is host-side tooling, out of the hot-path scope")

        def get_total_subsampling_factor(self):
            raise NotImplementedError("get_total_subsampling_factor method is not implemented")

        def encode(self, feat):
            raise NotImplementedError("encode method is not implemented")

        def scorers(self):
            raise NotImplementedError("decoders method is not implemented")
