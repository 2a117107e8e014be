"""Log-mel frontend on the device: waveform -> STFT -> power spectrum -> mel filterbank -> log.

reference: espnet2/asr/frontend/default.py:18-133 (DefaultFrontend with frontend_conf's WPE / beamformer switched
off, its default), espnet2/layers/stft.py:16-111 (Stft over torch.stft), espnet2/layers/log_mel.py:8-75 (LogMel over
librosa.filters.mel).  Same class names, constructor arguments and `forward(input, input_lengths) -> (feats, lens)`.

MI355X mapping: the DFT of all frames is ONE fp32 MFMA GEMM - the reflect-padded waveform is read in place as
overlapping rows (leading dimension = hop length), the window is folded into the [2F, n_fft] basis - followed by one
kernel that squares, applies the (short, triangular) mel filters, clamps, takes the log and zeroes the padded frames.
bf16 is never used here: the power spectrum spans > 100 dB.

librosa is a third-party dependency of the reference that is absent from this image (pinned `librosa>=0.8.0` in the
reference's setup.py); `mel_filterbank` restates its published `filters.mel` algorithm (Slaney / HTK mel scales, Slaney
area normalisation) and is checked against the two values printed in that function's docstring (the CPU test suite).
"""
import math

import numpy as np
import torch

from .. import ops
from ..nets.modules import make_pad_mask

try:  # pragma: no cover - only when the reference package is importable
    from espnet2.asr.frontend.abs_frontend import AbsFrontend  # type: ignore
except Exception:  # noqa: BLE001
    AbsFrontend = torch.nn.Module


def _hz_to_mel(f, htk):
    f = np.asarray(f, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m, htk):
    m = np.asarray(m, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk, norm='slaney') -> float32 [n_mels, 1 + n_fft//2]"""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin, htk), _hz_to_mel(fmax, htk), n_mels + 2), htk)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None].astype(np.float32)
    return weights


class Stft(torch.nn.Module):
    def __init__(self, n_fft=512, win_length=None, hop_length=128, window="hann", center=True, normalized=False,
                 onesided=True):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length, self.center, self.normalized, self.onesided = hop_length, center, normalized, onesided
        if window is not None and not hasattr(torch, f"{window}_window"):
            raise ValueError(f"{window} window is not implemented")
        self.window = window
        self._basis = None

    def extra_repr(self):
        return (f"n_fft={self.n_fft}, win_length={self.win_length}, hop_length={self.hop_length}, "
                f"center={self.center}, normalized={self.normalized}, onesided={self.onesided}")

    @property
    def n_freq(self):
        return self.n_fft // 2 + 1 if self.onesided else self.n_fft

    def basis(self, device):
        """[2F, n_fft]: rows 2j / 2j+1 = w[k] cos(2 pi j k / N) / -w[k] sin(2 pi j k / N) (window centred in n_fft)"""
        if self._basis is None or self._basis.device != torch.device(device):
            N, F = self.n_fft, self.n_freq
            w = np.ones(self.win_length) if self.window is None else \
                getattr(torch, f"{self.window}_window")(self.win_length, dtype=torch.float64).numpy()
            left = (N - self.win_length) // 2
            wf = np.zeros(N)
            wf[left:left + self.win_length] = w
            if self.normalized:
                wf = wf * N ** -0.5
            ang = 2.0 * math.pi * ((np.arange(F)[:, None] * np.arange(N)[None, :]) % N) / N
            b = np.empty((2 * F, N))
            b[0::2] = np.cos(ang) * wf
            b[1::2] = -np.sin(ang) * wf
            self._basis = torch.from_numpy(b.astype(np.float32)).to(device)
        return self._basis

    def frames(self, nsamples):
        pad = self.n_fft // 2 if self.center else 0
        return (nsamples + 2 * pad - self.n_fft) // self.hop_length + 1

    def spectrum(self, x):
        """x [B, L] -> (spec flat [B*rows_per_utt, 2F] interleaved (re, im), rows_per_utt, T)"""
        B, L = x.shape
        pad = self.n_fft // 2 if self.center else 0
        T = self.frames(L)
        if T < 1:
            raise ValueError("input is shorter than one STFT frame")
        rpu = -(-(L + 2 * pad) // self.hop_length)
        xp = ops.reflect_pad(x.contiguous().float(), pad, rpu * self.hop_length, self.n_fft)
        F2 = 2 * self.n_freq
        spec = torch.empty(B * rpu, F2, device=x.device, dtype=torch.float32)
        ops.gemm(xp, self.basis(x.device), spec, B * rpu, F2, self.n_fft, self.hop_length, self.n_fft, F2, precision=0)
        return spec, rpu, T

    def olens(self, ilens):
        """stft.py:104-108 (the length formula uses win_length where the framing uses n_fft)"""
        if self.center:
            ilens = ilens + 2 * (self.win_length // 2)
        return (ilens - self.win_length) // self.hop_length + 1

    def forward(self, input, ilens=None):
        """input (B, Nsamples) or (B, Nsamples, C) -> (B, Frames, [C,] Freq, 2), olens; padded frames zeroed"""
        bs = input.size(0)
        multi = input.dim() == 3
        if multi:
            input = input.transpose(1, 2).reshape(-1, input.size(1))
        spec, rpu, T = self.spectrum(input)
        F = self.n_freq
        out = spec.view(input.size(0), rpu, F, 2)[:, :T].contiguous()
        if multi:
            out = out.view(bs, -1, T, F, 2).transpose(1, 2).contiguous()
        if ilens is None:
            return out, None
        olens = self.olens(torch.as_tensor(ilens))
        keep = (~make_pad_mask(olens.tolist(), T)).to(torch.uint8)              # [B, T]
        if multi:
            keep = keep.unsqueeze(-1).expand(bs, T, out.size(2))
        rows = out.view(-1, (out.numel() // keep.numel()))
        out = ops.mask_rows(rows, ops.h2d_cached("stft_keep", keep.contiguous().numpy().reshape(-1), out.device)).view(out.shape)
        return out, olens


class LogMel(torch.nn.Module):
    def __init__(self, fs=16000, n_fft=512, n_mels=80, fmin=None, fmax=None, htk=False, log_base=None):
        super().__init__()
        fmin = 0 if fmin is None else fmin
        fmax = fs / 2 if fmax is None else fmax
        self.mel_options = dict(sr=fs, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk)
        self.log_base = log_base
        melmat = mel_filterbank(**self.mel_options)                              # (n_mels, F)
        self.register_buffer("melmat", torch.from_numpy(np.ascontiguousarray(melmat.T)).float())
        nz = melmat > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0)
        hi = np.where(nz.any(1), melmat.shape[1] - nz[:, ::-1].argmax(1), 0)
        self.register_buffer("_lo", torch.from_numpy(lo.astype(np.int32)), persistent=False)
        self.register_buffer("_hi", torch.from_numpy(hi.astype(np.int32)), persistent=False)

    def extra_repr(self):
        return ", ".join(f"{k}={v}" for k, v in self.mel_options.items())

    @property
    def log_scale(self):
        return 1.0 if self.log_base is None else 1.0 / math.log(self.log_base)

    def from_spectrum(self, spec, rpu, T, B, flens_dev):
        F = self.melmat.shape[0]
        return ops.logmel(spec, spec.shape[1], rpu, self.melmat, self._lo, self._hi, flens_dev, B, T, F, self.log_scale)

    def forward(self, feat, ilens=None):
        """feat (B, T, F) power spectrum -> (log-mel (B, T, n_mels), ilens)"""
        B, T, F = feat.shape
        fl = None
        if ilens is not None:
            fl = ops.h2d_cached("logmel_lens", np.asarray([int(v) for v in ilens], dtype=np.int32), feat.device)
        out = ops.logmel(feat.contiguous().float(), F, T, self.melmat, self._lo, self._hi, fl, B, T, F, self.log_scale,
                         power_input=True)
        if ilens is None:
            ilens = torch.full([B], T, dtype=torch.long)
        return out, ilens


class DefaultFrontend(AbsFrontend):
    """Stft -> power spectrum -> LogMel.  frontend_conf (WPE / MVDR beamformer of espnet/nets/pytorch_backend/frontends)
    is speech enhancement, outside this path: only its switched-off default is accepted."""

    def __init__(self, fs=16000, n_fft=512, win_length=None, hop_length=128, window="hann", center=True,
                 normalized=False, onesided=True, n_mels=80, fmin=None, fmax=None, htk=False, frontend_conf=None):
        super().__init__()
        if isinstance(fs, str):
            mult = {"k": 1000, "m": 1000000}.get(fs[-1].lower())
            fs = int(float(fs[:-1]) * mult) if mult else int(fs)
        if frontend_conf is not None and (frontend_conf.get("use_wpe") or frontend_conf.get("use_beamformer")):
            raise NotImplementedError("WPE / beamformer enhancement is outside the hot-path scope")
        self.stft = Stft(n_fft=n_fft, win_length=win_length, hop_length=hop_length, center=center, window=window,
                         normalized=normalized, onesided=onesided)
        self.frontend = None
        self.logmel = LogMel(fs=fs, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax, htk=htk)
        self.n_mels = n_mels

    def output_size(self):
        return self.n_mels

    def forward(self, input, input_lengths):
        if input.dim() == 3:      # default.py:107-115: one channel, random in training, the first otherwise
            ch = np.random.randint(input.size(2)) if self.training else 0
            input = input[:, :, ch]
        spec, rpu, T = self.stft.spectrum(input)
        feats_lens = self.stft.olens(torch.as_tensor(input_lengths).cpu())
        fl = ops.h2d_cached("frontend_lens", feats_lens.numpy().astype(np.int32), input.device)
        feats = self.logmel.from_spectrum(spec, rpu, T, input.size(0), fl)
        return feats, feats_lens
