"""Feature front-end on the HIP path: Kaldi-compatible log-mel filterbank and utterance CMVN.

Reference seam (paths relative to /root/reference/fairseq):
  * ``data/audio/audio_utils.py:59-79`` ``_get_torchaudio_fbank(waveform, sample_rate, n_bins=80)`` — waveform in int16
    range (``:31-32``), delegated to ``torchaudio.compliance.kaldi.fbank`` with its defaults.  torchaudio is third-party
    and absent from the reference tree and from this image: the kernel follows its published algorithm, parity unpinned.
  * ``data/audio/feature_transforms/utterance_cmvn.py`` ``UtteranceCMVN`` (``from_config_dict`` / ``__call__``), registered as
    ``utterance_cmvn`` (``feature_transforms/__init__.py:18-36``).

The reference runs both per utterance on dataloader CPU workers; here a whole batch of raw audio is featurised in two
launches (``s2t_fbank``, ``s2t_utterance_cmvn`` in ``csrc/frontend.hip``).  There is no CPU fallback.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import kernels as K

AUDIO_FEATURE_TRANSFORM_REGISTRY = {}


def register_audio_feature_transform(name):
    """feature_transforms/__init__.py:18-36."""

    def deco(cls):
        if name in AUDIO_FEATURE_TRANSFORM_REGISTRY:
            raise ValueError(f"Cannot register duplicate transform ({name})")
        AUDIO_FEATURE_TRANSFORM_REGISTRY[name] = cls
        return cls

    return deco


def get_audio_feature_transform(name):
    return AUDIO_FEATURE_TRANSFORM_REGISTRY[name]


_TABLES = {}


def _mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def kaldi_tables(sample_rate: int, n_bins: int, frame_length_ms=25.0, frame_shift_ms=10.0, low_freq=20.0, high_freq=0.0,
                 device="cuda"):
    """(win, shift, nfft, window[win], mel_t[nfft/2+1][n_bins]) of torchaudio.compliance.kaldi.fbank's defaults:
    povey window = hann(win, periodic=False)**0.85; triangular mel filters between low_freq and Nyquist + high_freq on
    mel(f) = 1127 ln(1 + f/700), Nyquist column zero.  Cached per configuration on the device."""
    key = (sample_rate, n_bins, frame_length_ms, frame_shift_ms, low_freq, high_freq, str(device))
    if key in _TABLES:
        return _TABLES[key]
    win = int(sample_rate * frame_length_ms * 0.001)
    shift = int(sample_rate * frame_shift_ms * 0.001)
    nfft = 1 << (win - 1).bit_length()
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win, dtype=np.float64) / (win - 1))) ** 0.85
    nyq = 0.5 * sample_rate
    hi = high_freq + nyq if high_freq <= 0.0 else high_freq
    mlo, mhi = _mel_scale(low_freq), _mel_scale(hi)
    delta = (mhi - mlo) / (n_bins + 1)
    b = np.arange(n_bins, dtype=np.float64)[None, :]
    left, center, right = mlo + b * delta, mlo + (b + 1.0) * delta, mlo + (b + 2.0) * delta
    m = _mel_scale(sample_rate / nfft * np.arange(nfft // 2, dtype=np.float64))[:, None]
    banks = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    mel_t = np.concatenate([banks, np.zeros((1, n_bins))], 0)  # [nfft/2+1][n_bins]
    out = (win, shift, nfft, torch.from_numpy(window.astype(np.float32)).to(device),
           torch.from_numpy(mel_t.astype(np.float32)).contiguous().to(device))
    _TABLES[key] = out
    return out


def fbank_batch(waveforms: Union[torch.Tensor, Sequence[torch.Tensor]], n_samples: Optional[torch.Tensor] = None,
                sample_rate: int = 16000, n_bins: int = 80) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched ``_get_torchaudio_fbank``: waveforms in int16 range, either a padded ``(B, N)`` CUDA tensor with
    ``n_samples (B,)`` or a list of 1-D tensors.  Returns ``(features (B, T, n_bins) fp32, n_frames (B,) int32)`` with
    frames past an utterance's end zero-filled (the collater's padding, speech_to_text_dataset.py:267-285)."""
    if not isinstance(waveforms, torch.Tensor):
        lens = [int(w.numel()) for w in waveforms]
        dev = waveforms[0].device
        wav = torch.zeros(len(lens), max(lens), dtype=torch.float32, device=dev)
        for i, w in enumerate(waveforms):
            wav[i, :lens[i]].copy_(w.reshape(-1))
        n_samples = torch.tensor(lens, dtype=torch.int32, device=dev)
        waveforms = wav
    if not waveforms.is_cuda:
        raise RuntimeError("s2t_amd.audio runs on the GPU only; there is no CPU fallback")
    wav = waveforms.float().contiguous()
    n_samples = n_samples.to(torch.int32).to(wav.device)
    win, shift, nfft, window, mel_t = kaldi_tables(sample_rate, n_bins, device=wav.device)
    n_frames = torch.where(n_samples >= win, 1 + torch.div(n_samples - win, shift, rounding_mode="floor"),
                           torch.zeros_like(n_samples)).to(torch.int32)
    N = wav.shape[1]
    max_frames = 1 + (N - win) // shift if N >= win else 0
    feat = torch.empty(wav.shape[0], max_frames, n_bins, dtype=torch.float32, device=wav.device)
    if max_frames > 0:
        K.fbank(wav, n_samples, feat, max_frames, win, shift, nfft, window, mel_t)
    return feat, n_frames


def get_torchaudio_fbank(waveform, sample_rate, n_bins=80):
    """Single-utterance mirror of audio_utils.py:59-79: waveform (N,) or (1, N) in int16 range -> (T, n_bins).
    numpy in -> numpy out (through the GPU), tensor in -> tensor out."""
    as_numpy = isinstance(waveform, np.ndarray)
    w = torch.from_numpy(waveform) if as_numpy else waveform
    w = w.reshape(-1).float().cuda()
    feat, n = fbank_batch([w], sample_rate=sample_rate, n_bins=n_bins)
    out = feat[0, :int(n[0])]
    return out.cpu().numpy() if as_numpy else out


@register_audio_feature_transform("utterance_cmvn")
class UtteranceCMVN:
    """data/audio/feature_transforms/utterance_cmvn.py — per-utterance (x - mean) / sqrt(max(E[x^2] - mean^2, 1e-10))
    over the time axis."""

    @classmethod
    def from_config_dict(cls, config: Optional[Dict] = None):
        _config = {} if config is None else config
        return UtteranceCMVN(_config.get("norm_means", True), _config.get("norm_vars", True),
                             _config.get("cmvn_no_axis", False))

    def __init__(self, norm_means=True, norm_vars=True, no_axis=False):
        if no_axis:
            raise NotImplementedError("cmvn_no_axis (statistics over both axes) is not used by the recipes")
        self.norm_means, self.norm_vars, self.no_axis = norm_means, norm_vars, no_axis

    def __repr__(self):
        return self.__class__.__name__ + f"(norm_means={self.norm_means}, norm_vars={self.norm_vars}, no_axis={self.no_axis})"

    def apply_batch(self, feat: torch.Tensor, n_frames: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat (B, T, C) fp32 CUDA, statistics over each utterance's first n_frames rows; padded rows are left as they
        are (zeros)."""
        if not feat.is_cuda:
            raise RuntimeError("s2t_amd.audio runs on the GPU only; there is no CPU fallback")
        feat = feat.contiguous()
        out = feat.clone() if out is None else out
        K.utterance_cmvn(feat, out, n_frames.to(torch.int32), self.norm_means, self.norm_vars)
        return out

    def __call__(self, x):
        as_numpy = isinstance(x, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x.float()
        t = t.cuda().unsqueeze(0)
        n = torch.tensor([t.shape[1]], dtype=torch.int32, device=t.device)
        y = self.apply_batch(t, n)[0]
        return y.cpu().numpy() if as_numpy else y
