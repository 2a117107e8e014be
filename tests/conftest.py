import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def split_golden(g, dtype=torch.float32):
    """-> (plain arrays as tensors, state_dict, grads)"""
    plain, sd, grads = {}, {}, {}
    for k, v in g.items():
        t = torch.from_numpy(np.asarray(v))
        if t.is_floating_point():
            t = t.to(dtype)
        if k.startswith("sd/"):
            sd[k[3:]] = t
        elif k.startswith("grad/"):
            grads[k[5:]] = t
        elif k.startswith("sd_after/"):
            plain[k] = t
        else:
            plain[k] = t
    return plain, sd, grads


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import asr_oracle
    return asr_oracle


def has_gpu():
    return torch.cuda.is_available()


def seeded_weights():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded_weights as SW
    return SW


POSTNORM_VARIANTS = (("post", False, False), ("postcat", False, True), ("precat", True, True))


def postnorm_layer(kind, normalize_before, concat_after):
    """our conformer / transformer encoder layer or decoder layer (size 64, 4 heads, 96 units) with the name-keyed weights
    oracle/gen_golden_r4b.py gave the reference layers of postnorm_layers.npz"""
    from espnet_amd.nets import modules as M
    D, H, U = 64, 4, 96
    SW = seeded_weights()
    if kind == "conf":
        m = M.ConformerEncoderLayer(D, M.RelPositionMultiHeadedAttention(H, D, 0.0), M.PositionwiseFeedForward(D, U, 0.0, "swish"),
                                    M.PositionwiseFeedForward(D, U, 0.0, "swish"), M.ConvolutionModule(D, 7, "swish"), 0.0,
                                    normalize_before, concat_after)
        return SW.fill_parameters(m, salt=990)
    if kind == "trf":
        m = M.TransformerEncoderLayer(D, M.MultiHeadedAttention(H, D, 0.0), M.PositionwiseFeedForward(D, U, 0.0), 0.0,
                                      normalize_before, concat_after)
        return SW.fill_parameters(m, salt=991)
    m = M.DecoderLayer(D, M.MultiHeadedAttention(H, D, 0.0), M.MultiHeadedAttention(H, D, 0.0), M.PositionwiseFeedForward(D, U, 0.0), 0.0,
                       normalize_before, concat_after)
    return SW.fill_parameters(m, salt=992)


def e2e_d512_model(dropout=0.0):
    """our espnet1 Conformer E2E at the width of the reference's large recipes (adim 512, aheads 8, eunits = dunits = 2048) with
    the name-keyed weights oracle/gen_golden_r4b.py gave the reference model of e2e_conformer_d512.npz; -> (model on CPU, oracle cfg)"""
    import argparse
    from espnet_amd.nets.e2e_asr_conformer import E2E
    ns = argparse.Namespace(
        adim=512, aheads=8, elayers=2, eunits=2048, dlayers=1, dunits=2048, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=dropout,
        transformer_length_normalized_loss=False, transformer_encoder_pos_enc_layer_type="rel_pos",
        transformer_encoder_selfattn_layer_type="rel_selfattn", macaron_style=True, use_cnn_module=True,
        cnn_module_kernel=31)
    model = seeded_weights().fill_parameters(E2E(80, 50, ns), salt=512)
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=8, mtlalpha=0.3, lsm_weight=0.1, odim=50)
    return model, cfg


def e2e_dk64_model(dropout=0.0):
    """our espnet1 Conformer E2E at adim 256 / aheads 4 (d_k = 64: the width at which bf16 mode dispatches the fused
    attention kernels) with the name-keyed weights oracle/gen_golden_r2.py gave the reference model of
    e2e_conformer_dk64.npz; -> (model on CPU, oracle cfg)"""
    import argparse
    from espnet_amd.nets.e2e_asr_conformer import E2E
    ns = argparse.Namespace(
        adim=256, aheads=4, elayers=2, eunits=64, dlayers=1, dunits=64, mtlalpha=0.3, lsm_weight=0.1, dropout_rate=dropout,
        transformer_length_normalized_loss=False, transformer_encoder_pos_enc_layer_type="rel_pos",
        transformer_encoder_selfattn_layer_type="rel_selfattn", macaron_style=True, use_cnn_module=True,
        cnn_module_kernel=31)
    model = seeded_weights().fill_parameters(E2E(20, 50, ns), salt=5)
    cfg = dict(conformer=True, rel_pos=True, activation="swish", aheads=4, mtlalpha=0.3, lsm_weight=0.1, odim=50)
    return model, cfg
