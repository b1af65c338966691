"""The steps either side of the model in the reference's N-DNS denoising loop, on the device.

Reference: ``sparseRNNs/train_helpers.py:1382-1412`` (``stft_splitter`` / ``stft_mixer``: ``jax.scipy.signal.stft`` /
``istft`` with nperseg = nfft = 512, hop 128, boxcar window, one-sided, the SciPy defaults ``boundary="zeros"``,
``padded=True``, ``scaling="spectrum"``), ``:15-53`` (``si_snr_jax``) and ``sparseRNNs/fxprun.py:63-88`` (the
validation step that strings them together around ``model(fxp_x)``).

Everything here is plain torch tensor code (rocFFT does the transforms on a ROCm device), so the whole
STFT -> fixed-point model -> iSTFT -> SI-SNR chain stays in HBM.  ``jax.scipy.signal`` mirrors ``scipy.signal``, which is
importable here: ``tests/test_cpu_suite.py`` checks these functions against it.
"""
from __future__ import annotations

import torch

NFFT = 512
HOP = 128
STFT_MAG_MEAN = 0.0007  # fxprun.py:65


def _frames(x: torch.Tensor) -> torch.Tensor:
    """scipy.signal.stft's segmentation: 256 zeros each side (boundary="zeros"), zero padding at the end to a whole
    number of hops (padded=True), then windows of 512 every 128 samples.  (..., T) -> (..., n_seg, 512)"""
    x = torch.nn.functional.pad(x, (NFFT // 2, NFFT // 2))
    nadd = (-(x.shape[-1] - NFFT) % HOP) % NFFT
    if nadd:
        x = torch.nn.functional.pad(x, (0, nadd))
    return x.unfold(-1, NFFT, HOP)


def stft(audio: torch.Tensor) -> torch.Tensor:
    """Complex one-sided STFT, (..., T) -> (..., 257, n_seg), scaled by 1/sum(window) = 1/512 ("spectrum")."""
    seg = _frames(audio.to(torch.float32))
    z = torch.fft.rfft(seg, n=NFFT, dim=-1) / NFFT
    return z.transpose(-1, -2)


def stft_splitter(audio: torch.Tensor):
    """train_helpers.py:1382-1396: magnitude and phase, each (..., 257, n_seg)."""
    z = stft(audio)
    return z.abs(), z.angle()


def istft(z: torch.Tensor) -> torch.Tensor:
    """scipy.signal.istft for the same parameters: (..., 257, n_seg) -> (..., (n_seg - 1) * 128)."""
    seg = torch.fft.irfft(z.transpose(-1, -2), n=NFFT, dim=-1) * NFFT  # (..., n_seg, 512), undo the spectrum scaling
    n_seg = seg.shape[-2]
    total = NFFT + HOP * (n_seg - 1)
    lead = seg.shape[:-2]
    flat = seg.reshape(-1, n_seg, NFFT)
    # overlap-add; with the boxcar window the normaliser is the number of windows that cover a sample
    idx = (torch.arange(n_seg, device=seg.device)[:, None] * HOP + torch.arange(NFFT, device=seg.device)[None, :]).reshape(-1)
    out = torch.zeros(flat.shape[0], total, dtype=seg.dtype, device=seg.device)
    out.index_add_(1, idx, flat.reshape(flat.shape[0], -1))
    norm = torch.zeros(total, dtype=seg.dtype, device=seg.device)
    norm.index_add_(0, idx, torch.ones_like(idx, dtype=seg.dtype))
    out = out / torch.where(norm > 1e-10, norm, torch.ones_like(norm))
    return out[:, NFFT // 2: total - NFFT // 2].reshape(*lead, -1)


def stft_mixer(stft_mag: torch.Tensor, stft_angle: torch.Tensor) -> torch.Tensor:
    """train_helpers.py:1399-1412."""
    return istft(torch.polar(stft_mag.to(torch.float32), stft_angle.to(torch.float32)))


def si_snr(target: torch.Tensor, estimate: torch.Tensor) -> torch.Tensor:
    """train_helpers.py:15-53 (last dimension = time)."""
    eps = 1e-8
    s_t = target - target.mean(dim=-1, keepdim=True)
    s_e = estimate - estimate.mean(dim=-1, keepdim=True)
    dot = (s_t * s_e).sum(dim=-1, keepdim=True)
    proj = dot * s_t / (s_t ** 2).sum(dim=-1, keepdim=True)
    noise = s_e - proj
    sdr = (proj ** 2).sum(dim=-1) / ((noise ** 2).sum(dim=-1) + eps)
    return 10.0 * torch.log10(sdr + eps)


def denoise(model, inp_bits: int, inp_exp: int, noisy: torch.Tensor):
    """fxprun.py:63-78: noisy audio (B, T) -> (cleaned audio, cleaned magnitude, noisy magnitude).

    ``model`` is an ``FxpRegressionModel`` (or anything callable FxpArray -> FxpArray with ``to_float``)."""
    from .fxparray import RoundingMode, fxp_from_fp

    mag, phase = stft_splitter(noisy)
    x = (mag - STFT_MAG_MEAN).transpose(-1, -2).contiguous()  # (B, n_seg, 257)
    fx = fxp_from_fp(x, bits=inp_bits, exp=inp_exp, signed=True, round_mode=RoundingMode.FLOOR)
    mask = model(fx).to_float().transpose(-1, -2)
    cleaned_mag = mag * (1.0 + mask)
    return stft_mixer(cleaned_mag, phase), cleaned_mag, mag
