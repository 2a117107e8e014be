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
