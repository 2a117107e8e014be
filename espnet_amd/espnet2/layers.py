"""espnet2 feature-side layers on the espnet_amd HIP kernels: SpecAug (TimeWarp, MaskAlongAxis), GlobalMVN,
UtteranceMVN.  Same class names / constructor arguments / call signatures as the reference
(espnet2/asr/specaug/specaug.py, espnet2/layers/{time_warp,mask_along_axis,global_mvn,utterance_mvn}.py).

Random draws follow the reference call for call (torch.randint with the same bounds and shapes, in the same
order) on the HOST generator: a few integers per batch; the arithmetic over the [B,T,F] batch runs in one kernel.
"""
from pathlib import Path

import numpy as np
import torch

from .. import ops


def _lens_list(x_lengths, B, T):
    if x_lengths is None:
        return [T] * B
    return [int(v) for v in (x_lengths.tolist() if torch.is_tensor(x_lengths) else x_lengths)]


def _dev_i32(a, device):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)


try:  # pragma: no cover - only when the reference package is importable: its registries type-check these bases
    from espnet2.asr.specaug.abs_specaug import AbsSpecAug  # type: ignore
    from espnet2.layers.abs_normalize import AbsNormalize  # type: ignore
except Exception:  # noqa: BLE001
    AbsSpecAug = AbsNormalize = torch.nn.Module


class TimeWarp(torch.nn.Module):
    """reference: espnet2/layers/time_warp.py:56-94 (draws only; the interpolation runs inside SpecAug's kernel)"""

    def __init__(self, window=80, mode="bicubic"):
        super().__init__()
        if mode != "bicubic":
            raise NotImplementedError("time_warp mode %r: bicubic (the reference default) is on the HIP path" % mode)
        self.window, self.mode = window, mode

    def extra_repr(self):
        return f"window={self.window}, mode={self.mode}"

    def _draw_one(self, t):
        """time_warp.py:34-38: None when the utterance is too short to warp"""
        w = self.window
        if t - w <= w:
            return None
        center = int(torch.randint(w, t - w, (1,))[0])
        warped = int(torch.randint(center - w, center + w, (1,))[0]) + 1
        return center, warped

    def draw(self, lens, T):
        """-> (center[B], warped[B], out_lens, out_T); center = -1 where no warp applies"""
        B = len(lens)
        if all(le == lens[0] for le in lens):
            d = self._draw_one(T)            # one draw for the whole batch, over the padded length (time_warp.py:76-78)
            c, w = d if d is not None else (-1, 0)
            return [c] * B, [w] * B, None, T
        cs, ws = [], []
        for le in lens:                      # per-utterance path (time_warp.py:81-90): own length, own draw, re-padded
            d = self._draw_one(le)
            c, w = d if d is not None else (-1, 0)
            cs.append(c)
            ws.append(w)
        return cs, ws, lens, max(lens)

    def forward(self, x, x_lengths=None):
        B, T, F = x.shape
        lens = _lens_list(x_lengths, B, T)
        c, w, out_lens, out_T = self.draw(lens, T)
        xin = x if out_T == T else x[:, :out_T].contiguous()
        y = ops.specaug(xin.contiguous(), _dev_i32(out_lens, x.device) if out_lens is not None else None,
                        _dev_i32(c, x.device), _dev_i32(w, x.device))
        return y, x_lengths


class MaskAlongAxis(torch.nn.Module):
    """reference: espnet2/layers/mask_along_axis.py:65-128 (replace_with_zero=True)"""

    def __init__(self, mask_width_range=(0, 30), num_mask=2, dim="time", replace_with_zero=True):
        super().__init__()
        if isinstance(mask_width_range, int):
            mask_width_range = (0, mask_width_range)
        if len(mask_width_range) != 2:
            raise TypeError(f"mask_width_range must be a tuple of int and int values: {mask_width_range}")
        assert mask_width_range[1] > mask_width_range[0]
        if isinstance(dim, str):
            if dim == "time":
                dim = 1
            elif dim == "freq":
                dim = 2
            else:
                raise ValueError("dim must be int, 'time' or 'freq'")
        if not replace_with_zero:
            raise NotImplementedError("replace_with_zero=False (mean fill) is not on the HIP path")
        self.mask_axis = {1: "time", 2: "freq"}.get(dim, "unknown")
        self.mask_width_range, self.num_mask, self.dim, self.replace_with_zero = mask_width_range, num_mask, dim, True

    def extra_repr(self):
        return f"mask_width_range={self.mask_width_range}, num_mask={self.num_mask}, axis={self.mask_axis}"

    def draw(self, B, D):
        """mask_along_axis.py:33-44: widths then positions, both (B, num_mask)"""
        length = torch.randint(self.mask_width_range[0], self.mask_width_range[1], (B, self.num_mask))
        pos = torch.randint(0, max(1, D - int(length.max())), (B, self.num_mask))
        return pos.numpy(), length.numpy()

    def forward(self, spec, spec_lengths=None):
        B, T, F = spec.shape
        pos, length = self.draw(B, spec.shape[self.dim])
        kw = dict(tpos=_dev_i32(pos, spec.device), tlen=_dev_i32(length, spec.device)) if self.dim == 1 else \
            dict(fpos=_dev_i32(pos, spec.device), flen=_dev_i32(length, spec.device))
        return ops.specaug(spec.contiguous(), **kw), spec_lengths


class SpecAug(AbsSpecAug):
    """reference: espnet2/asr/specaug/specaug.py:19-84.  time warp -> frequency masks -> time masks, fused into ONE
    kernel launch (the draws are made in the reference's order first)."""

    def __init__(self, apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True,
                 freq_mask_width_range=(0, 20), num_freq_mask=2, apply_time_mask=True,
                 time_mask_width_range=(0, 100), num_time_mask=2):
        if not apply_time_warp and not apply_time_mask and not apply_freq_mask:
            raise ValueError("Either one of time_warp, time_mask, or freq_mask should be applied")
        super().__init__()
        self.apply_time_warp, self.apply_freq_mask, self.apply_time_mask = apply_time_warp, apply_freq_mask, apply_time_mask
        self.time_warp = TimeWarp(window=time_warp_window, mode=time_warp_mode) if apply_time_warp else None
        self.freq_mask = MaskAlongAxis(dim="freq", mask_width_range=freq_mask_width_range,
                                       num_mask=num_freq_mask) if apply_freq_mask else None
        self.time_mask = MaskAlongAxis(dim="time", mask_width_range=time_mask_width_range,
                                       num_mask=num_time_mask) if apply_time_mask else None

    def forward(self, x, x_lengths=None):
        B, T, F = x.shape
        dev = x.device
        kw, out_T = {}, T
        if self.time_warp is not None:
            c, w, out_lens, out_T = self.time_warp.draw(_lens_list(x_lengths, B, T), T)
            kw.update(center=_dev_i32(c, dev), warped=_dev_i32(w, dev))
            if out_lens is not None:
                kw["lens"] = _dev_i32(out_lens, dev)
        if self.freq_mask is not None:
            pos, length = self.freq_mask.draw(B, F)
            kw.update(fpos=_dev_i32(pos, dev), flen=_dev_i32(length, dev))
        if self.time_mask is not None:
            pos, length = self.time_mask.draw(B, out_T)
            kw.update(tpos=_dev_i32(pos, dev), tlen=_dev_i32(length, dev))
        xin = x if out_T == T else x[:, :out_T]
        return ops.specaug(xin.contiguous(), **kw), x_lengths


class GlobalMVN(AbsNormalize):
    """reference: espnet2/layers/global_mvn.py:14-121 (stats from a Kaldi-style .npy or a count/sum/sum_square .npz)"""

    def __init__(self, stats_file, norm_means=True, norm_vars=True, eps=1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps
        self.stats_file = Path(stats_file)
        stats = np.load(self.stats_file)
        if isinstance(stats, np.ndarray):
            count = stats[0].flatten()[-1]
            mean = stats[0, :-1] / count
            var = stats[1, :-1] / count - mean * mean
        else:
            count = stats["count"]
            mean = stats["sum"] / count
            var = stats["sum_square"] / count - mean * mean
        std = np.sqrt(np.maximum(var, eps))
        self.register_buffer("mean", torch.from_numpy(mean))
        self.register_buffer("std", torch.from_numpy(std))

    def extra_repr(self):
        return f"stats_file={self.stats_file}, norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def forward(self, x, ilens=None):
        B, T, F = x.shape
        if ilens is None:
            ilens = x.new_full([B], T)
        if self.mean.dtype != torch.float32 or self.mean.device != x.device:
            self.mean = self.mean.to(x.device, torch.float32)
            self.std = self.std.to(x.device, torch.float32)
        lens = _dev_i32(_lens_list(ilens, B, T), x.device)
        y = ops.global_mvn(x.contiguous(), lens, self.mean if self.norm_means else None,
                           self.std if self.norm_vars else None)
        return y, ilens


class UtteranceMVN(AbsNormalize):
    """reference: espnet2/layers/utterance_mvn.py:9-88"""

    def __init__(self, norm_means=True, norm_vars=False, eps=1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps

    def extra_repr(self):
        return f"norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def forward(self, x, ilens=None):
        B, T, F = x.shape
        if ilens is None:
            ilens = x.new_full([B], T)
        lens = _dev_i32(_lens_list(ilens, B, T), x.device)
        return ops.utterance_mvn(x.contiguous(), lens, self.norm_means, self.norm_vars, self.eps), ilens
