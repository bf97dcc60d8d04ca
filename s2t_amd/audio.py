"""Feature front-end on the HIP path: Kaldi-compatible log-mel filterbank and utterance CMVN.

Reference seam (paths relative to /root/reference/fairseq):
  * ``data/audio/audio_utils.py:59-79`` ``_get_torchaudio_fbank(waveform, sample_rate, n_bins=80)`` — waveform in int16
    range (``:31-32``), delegated to ``torchaudio.compliance.kaldi.fbank`` with its defaults.  torchaudio is third-party
    and absent from the reference tree and from this image: the kernel follows its published algorithm; parity is unpinned against
    the reference and cross-checked against Hugging Face's independent port of the same call (tests/golden/fbank_hf_speech2text.npz).
  * ``data/audio/feature_transforms/utterance_cmvn.py`` ``UtteranceCMVN`` (``from_config_dict`` / ``__call__``), registered as
    ``utterance_cmvn`` (``feature_transforms/__init__.py:18-36``).

The reference runs both per utterance on dataloader CPU workers; here a whole batch of raw audio is featurised in two
launches (``s2t_fbank``, ``s2t_utterance_cmvn`` in ``csrc/frontend.hip``).  There is no CPU fallback.
"""
import math
from typing import Dict, Optional, Sequence, Tuple, Union

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


@register_audio_feature_transform("global_cmvn")
class GlobalCMVN:
    """data/audio/feature_transforms/global_cmvn.py — (x - mean) / std with the per-bin statistics of a ``.npz``
    (keys ``mean``, ``std``).  One per-column affine launch (``s2t_bn_act_fwd`` with the identity activation:
    y = x * (1/std) + (-mean/std))."""

    @classmethod
    def from_config_dict(cls, config: Optional[Dict] = None):
        _config = {} if config is None else config
        return GlobalCMVN(_config.get("stats_npz_path"))

    def __init__(self, stats_npz_path):
        self.stats_npz_path = stats_npz_path
        stats = np.load(stats_npz_path)
        self.mean, self.std = stats["mean"], stats["std"]
        self._dev = None

    def __repr__(self):
        return self.__class__.__name__ + f'(stats_npz_path="{self.stats_npz_path}")'

    def apply_batch(self, feat: torch.Tensor, n_frames: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat (B, T, C) fp32 CUDA; rows at or beyond n_frames[b] come out zero (the collater's padding)."""
        if not feat.is_cuda:
            raise RuntimeError("s2t_amd.audio runs on the GPU only; there is no CPU fallback")
        if self._dev is None or self._dev[0].device != feat.device:
            inv = 1.0 / np.asarray(self.std, dtype=np.float64)
            self._dev = (torch.tensor(inv, dtype=torch.float32, device=feat.device),
                         torch.tensor(-np.asarray(self.mean, dtype=np.float64) * inv, dtype=torch.float32, device=feat.device))
        feat = feat.contiguous()
        B, T, C = feat.shape
        out = torch.empty_like(feat)
        lens = None if n_frames is None else n_frames.to(device=feat.device, dtype=torch.int32)
        K.bn_act_fwd(feat, out, self._dev[0], self._dev[1], "none", B * T, C, lens, T)
        return out

    def __call__(self, x):
        as_numpy = isinstance(x, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x.float()
        y = self.apply_batch(t.cuda().unsqueeze(0))[0]
        return y.cpu().numpy() if as_numpy else y


class CompositeAudioFeatureTransform:
    """feature_transforms/__init__.py:51-82 — the configured transforms applied in order."""

    @classmethod
    def from_config_dict(cls, config=None):
        _config = {} if config is None else config
        _transforms = _config.get("transforms")
        if _transforms is None:
            return None
        return CompositeAudioFeatureTransform(
            [get_audio_feature_transform(t).from_config_dict(_config.get(t)) for t in _transforms])

    def __init__(self, transforms):
        self.transforms = [t for t in transforms if t is not None]

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x

    def __repr__(self):
        return self.__class__.__name__ + "(" + ", ".join(repr(t) for t in self.transforms) + ")"


@register_audio_feature_transform("specaugment")
class SpecAugmentTransform:
    """data/audio/feature_transforms/specaugment.py — SpecAugment frequency and time masking.  The interval draws use
    ``numpy.random`` in exactly the reference's order (per mask: width, then start; frequency masks first), so a seeded
    run masks the same cells; the masking itself runs on the device for the whole batch (``s2t_specaugment``).
    ``time_warp_W > 0``: the split point and the shift are drawn first (specaugment.py:101-102), the two linear resizes
    along time run on the device (``s2t_time_warp``: OpenCV's INTER_LINEAR row mapping restated, cv2 itself is absent)."""

    @classmethod
    def from_config_dict(cls, config: Optional[Dict] = None):
        _config = {} if config is None else config
        return SpecAugmentTransform(_config.get("time_warp_W", 0), _config.get("freq_mask_N", 0), _config.get("freq_mask_F", 0),
                                    _config.get("time_mask_N", 0), _config.get("time_mask_T", 0),
                                    _config.get("time_mask_p", 0.0), _config.get("mask_value", None))

    def __init__(self, time_warp_w=0, freq_mask_n=0, freq_mask_f=0, time_mask_n=0, time_mask_t=0, time_mask_p=0.0,
                 mask_value=0.0):
        if freq_mask_n > 0:
            assert freq_mask_f > 0, f"freq_mask_F ({freq_mask_f}) must be larger than 0 when doing freq masking."
        if time_mask_n > 0:
            assert time_mask_t > 0, f"time_mask_T ({time_mask_t}) must be larger than 0 when doing time masking."
        self.time_warp_w, self.freq_mask_n, self.freq_mask_f = time_warp_w, freq_mask_n, freq_mask_f
        self.time_mask_n, self.time_mask_t, self.time_mask_p, self.mask_value = time_mask_n, time_mask_t, time_mask_p, mask_value

    def __repr__(self):
        return (self.__class__.__name__ + f"(time_warp_w={self.time_warp_w}, freq_mask_n={self.freq_mask_n}, "
                f"freq_mask_f={self.freq_mask_f}, time_mask_n={self.time_mask_n}, time_mask_t={self.time_mask_t}, "
                f"time_mask_p={self.time_mask_p})")

    def draw_warp(self, num_frames: int, num_freqs: int):
        """(w0, w) of the time warp for one utterance, (0, 0) when none applies — drawn before the masks."""
        if num_frames == 0 or num_freqs < self.freq_mask_f or self.time_warp_w <= 0 or 2 * self.time_warp_w >= num_frames:
            return (0, 0)
        w0 = np.random.randint(self.time_warp_w, num_frames - self.time_warp_w)
        w = np.random.randint(-self.time_warp_w + 1, self.time_warp_w)
        return (int(w0), int(w))

    def draw(self, num_frames: int, num_freqs: int):
        """The reference's random draws for one utterance -> (freq intervals, time intervals) as (start, width) lists
        padded with empty intervals to freq_mask_n / time_mask_n entries."""
        fm = [(0, 0)] * self.freq_mask_n
        tm = [(0, 0)] * self.time_mask_n
        if num_frames == 0 or num_freqs < self.freq_mask_f:
            return fm, tm
        for i in range(self.freq_mask_n):
            f = np.random.randint(0, self.freq_mask_f)
            f0 = np.random.randint(0, num_freqs - f)
            fm[i] = (int(f0), int(f))
        max_t = min(self.time_mask_t, math.floor(num_frames * self.time_mask_p))
        if max_t < 1:
            return fm, tm
        for i in range(self.time_mask_n):
            t = np.random.randint(0, max_t)
            t0 = np.random.randint(0, num_frames - t)
            tm[i] = (int(t0), int(t))
        return fm, tm

    def apply_batch(self, feat: torch.Tensor, n_frames: torch.Tensor) -> torch.Tensor:
        """In place on feat (B, T, C) fp32 CUDA; utterances are drawn in batch order."""
        if not feat.is_cuda:
            raise RuntimeError("s2t_amd.audio runs on the GPU only; there is no CPU fallback")
        assert feat.is_contiguous() and feat.dtype == torch.float32
        B, T, Cf = feat.shape
        nf = n_frames.to(torch.int32)
        rows, warps = [], []
        for n in nf.tolist():
            warps.append(self.draw_warp(int(n), Cf))
            fm, tm = self.draw(int(n), Cf)
            rows.append(fm + tm)
        nm = self.freq_mask_n + self.time_mask_n
        use_mean = self.mask_value is None
        value = torch.full((B,), 0.0 if use_mean else float(self.mask_value), dtype=torch.float32, device=feat.device)
        nf_dev = nf.to(feat.device)
        if any(w0 > 0 for w0, _ in warps):
            warped = torch.empty_like(feat)
            # the fill value of mask_value = None is the mean of the UN-warped utterance (specaugment.py:89-90)
            K.time_warp(feat, warped, nf_dev, torch.tensor(warps, dtype=torch.int32).view(B, 2).to(feat.device),
                        mean_out=value if use_mean else None)
            feat.copy_(warped)
            use_mean = False
        if nm == 0:
            return feat
        masks = torch.tensor(rows, dtype=torch.int32).view(B, nm, 2).to(feat.device)
        K.specaugment(feat, nf_dev, masks, self.freq_mask_n, self.time_mask_n, value, use_mean)
        return feat

    def __call__(self, spectrogram):
        as_numpy = isinstance(spectrogram, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(spectrogram, dtype=np.float32)) if as_numpy else spectrogram.float()
        assert t.dim() == 2, "spectrogram must be a 2-D tensor."
        t = t.cuda().clone().unsqueeze(0).contiguous()
        n = torch.tensor([t.shape[1]], dtype=torch.int32)
        out = self.apply_batch(t, n)[0]
        return out.cpu().numpy() if as_numpy else out
