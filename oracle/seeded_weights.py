"""Deterministic, name-keyed parameter values and compact gradient probes  --  TEST INFRASTRUCTURE ONLY.

The d = 256 fixtures (the width at which bf16 mode dispatches the fused attention kernels, d_k = 64) would be
tens of megabytes if they carried their weights and every gradient like the small fixtures do.  Instead both
sides - oracle/gen_golden_r2.py running the REFERENCE modules, and the tests running ours - fill the parameters
from this name-keyed generator, and the fixture stores, per parameter, either the whole reference gradient
(small tensors) or two random projections of it (large ones): G r and l^T G for fixed seeded vectors l, r with
G viewed as [shape[0], -1].  A gradient matrix that differs from the reference's changes both projections.
"""
import math
import zlib

import numpy as np
import torch

FULL_GRAD_MAX = 4096      # parameters up to this many elements keep their whole gradient in the fixture


def _gen(name, salt):
    return torch.Generator().manual_seed(zlib.crc32(("%s|%d" % (name, salt)).encode()))


def seeded_value(name, shape, salt=0):
    """1-D '...weight' of a normalisation layer: U(0.5, 1.5); other 1-D tensors: U(-0.1, 0.1);
    matrices / convolution kernels: U(-a, a) with a = 1 / sqrt(fan_in)."""
    g = _gen(name, salt)
    shape = tuple(shape)
    if len(shape) <= 1:
        u = torch.rand(shape, generator=g)
        return 0.5 + u if name.endswith("weight") else 0.2 * u - 0.1
    fan_in = int(np.prod(shape[1:]))
    a = 1.0 / math.sqrt(fan_in)
    return (2.0 * torch.rand(shape, generator=g) - 1.0) * a


def fill_parameters(module, salt=0):
    """in-place, parameters only (buffers such as BatchNorm running statistics keep their defaults)"""
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(seeded_value(name, p.shape, salt).to(p.device, p.dtype))
    return module


def probe_vectors(name, shape):
    rows = int(shape[0])
    cols = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    g = _gen(name, 977)
    return torch.randn(rows, generator=g, dtype=torch.float64), torch.randn(cols, generator=g, dtype=torch.float64)


def grad_record(name, grad):
    """-> dict of arrays to store for this parameter's gradient"""
    grad = grad.detach().cpu()
    if grad.numel() <= FULL_GRAD_MAX:
        return {"grad/" + name: grad.numpy()}
    l, r = probe_vectors(name, grad.shape)
    G = grad.double().reshape(grad.shape[0], -1)
    return {"gprobe_r/" + name: (G @ r).numpy(), "gprobe_l/" + name: (l @ G).numpy(),
            "gnorm/" + name: np.asarray(float(G.norm()))}


def ref_norm(name, fixture):
    if "grad/" + name in fixture:
        return float(np.linalg.norm(np.asarray(fixture["grad/" + name], dtype=np.float64)))
    return float(fixture["gnorm/" + name])


def grad_check(name, grad, fixture):
    """-> (kind, rel_err) of our gradient against what the fixture holds for this parameter"""
    grad = grad.detach().cpu().double()
    if "grad/" + name in fixture:
        ref = torch.from_numpy(np.asarray(fixture["grad/" + name])).double()
        return "full", float((grad - ref).norm() / (ref.norm() + 1e-30))
    l, r = probe_vectors(name, grad.shape)
    G = grad.reshape(grad.shape[0], -1)
    pr = torch.from_numpy(np.asarray(fixture["gprobe_r/" + name])).double()
    pl = torch.from_numpy(np.asarray(fixture["gprobe_l/" + name])).double()
    gn = float(fixture["gnorm/" + name])
    # E |dG r|^2 = |dG|_F^2 for r ~ N(0, I): the probe difference over |G|_F estimates the relative Frobenius error
    er = float((G @ r - pr).norm() / (gn + 1e-30))
    el = float((l @ G - pl).norm() / (gn + 1e-30))
    en = abs(float(G.norm()) - gn) / (gn + 1e-30)
    return "probe", max(er, el, en)


# ---- round 4: the decode fixture at BASELINE config 2's width (oracle/gen_golden_r4.py and the tests share this) ----
DECODE_R4 = dict(
    idim=80, odim=5000, salt=4, seed=4, lens=(1000, 640, 300), beam=10,
    # random weights give near-uniform posteriors (nothing ever ends, no CTC blank): the two output layers are sharpened and
    # <eos> / blank get a bias so that hypotheses end at different lengths and end detection fires, as on a trained model
    out_scale=4.0, eos_bias=7.0, blank_bias=12.0,
    ns=dict(adim=256, aheads=4, elayers=2, eunits=2048, dlayers=2, dunits=2048, mtlalpha=0.3, lsm_weight=0.1,
            dropout_rate=0.0, transformer_attn_dropout_rate=0.0, transformer_length_normalized_loss=False,
            transformer_init="pytorch", transformer_input_layer="conv2d", ctc_type="builtin", report_cer=False,
            report_wer=False, char_list=None, sym_space="<space>", sym_blank="<blank>",
            transformer_encoder_pos_enc_layer_type="rel_pos", transformer_encoder_selfattn_layer_type="rel_selfattn",
            transformer_encoder_activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=31))


def decode_r4_model(E2E, spec=DECODE_R4):
    """E2E = the Conformer E2E class of either side (reference or espnet_amd): same state_dict names -> same weights"""
    import argparse
    model = fill_parameters(E2E(spec["idim"], spec["odim"], argparse.Namespace(**spec["ns"])), salt=spec["salt"])
    with torch.no_grad():
        model.decoder.output_layer.weight *= spec["out_scale"]
        model.ctc.ctc_lo.weight *= spec["out_scale"]
        model.decoder.output_layer.bias[spec["odim"] - 1] += spec["eos_bias"]
        model.ctc.ctc_lo.bias[0] += spec["blank_bias"]
    return model.eval()


def decode_r4_inputs(spec=DECODE_R4):
    g = torch.Generator().manual_seed(spec["seed"])
    return [torch.randn(T, spec["idim"], generator=g) for T in spec["lens"]]


DECODE_R4_CASES = [(cw, ratio, 0.0 if ratio == 0.0 else 0.1) for cw in (0.0, 0.3, 1.0) for ratio in (0.0, 0.2)]
