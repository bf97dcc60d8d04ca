"""Feature front-end.

CPU: properties of the oracle's Kaldi-fbank restatement (a1 is third-party in the reference: torchaudio, absent here and
unpinned there, so the oracle cannot be pinned to the reference itself: it is anchored on the algorithm's own invariants and,
since round 6, cross-checked against an INDEPENDENT restatement of the same call — Hugging Face's Speech2TextFeatureExtractor,
the port of this very fairseq front-end, through tests/golden/fbank_hf_speech2text.npz, made by oracle/gen_golden_fbank.py) and
the CMVN oracle against the reference's formula.
GPU: ``s2t_fbank`` / ``s2t_utterance_cmvn`` through the C-ABI against the oracle and against that fixture."""
import numpy as np
import pytest
import torch

from oracle import s2t_oracle as O


def _tone(freq, n=16000, sr=16000, amp=0.3):
    t = np.arange(n) / sr
    return amp * np.sin(2 * np.pi * freq * t) * 32768.0


def test_oracle_fbank_frame_count_and_shapes():
    # snip_edges: 1 + (N - 400) // 160 frames of 80 bins; shorter than one window -> no frames
    assert O.kaldi_fbank(np.zeros(399)).shape == (0, 80)
    assert O.kaldi_fbank(np.zeros(400)).shape == (1, 80)
    assert O.kaldi_fbank(np.zeros(16000)).shape == (98, 80)
    assert O.kaldi_fbank(np.zeros(16000 + 79)).shape == (98, 80)
    assert O.kaldi_fbank(np.zeros(16000 + 80)).shape == (99, 80)
    # silence -> log(float32 eps) everywhere
    np.testing.assert_allclose(O.kaldi_fbank(np.zeros(1000)), np.log(np.finfo(np.float32).eps))


def test_oracle_fbank_tone_lands_in_its_mel_bin_and_dc_is_removed():
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(8000.0)
    delta = (hi - lo) / 81
    for freq in (300.0, 1000.0, 3000.0, 6000.0):
        f = O.kaldi_fbank(_tone(freq))
        centre = lo + (f[20].argmax() + 1) * delta
        assert abs(centre - mel(freq)) <= delta, freq
    w = _tone(1000.0)
    np.testing.assert_allclose(O.kaldi_fbank(w + 5000.0), O.kaldi_fbank(w), atol=1e-6)  # remove_dc_offset
    # energies scale quadratically with the amplitude: +2 log 2 per doubling
    np.testing.assert_allclose(O.kaldi_fbank(2 * w)[:, 20:40], O.kaldi_fbank(w)[:, 20:40] + 2 * np.log(2.0), atol=1e-6)


def test_oracle_fbank_agrees_with_the_hf_speech2text_port(golden_dir):
    """Two independent restatements of torchaudio.compliance.kaldi.fbank(wave * 2**15, num_mel_bins=80, sample_frequency=16000)
    (fairseq/data/audio/audio_utils.py:59-79): the oracle's float64 numpy and transformers' float32 numpy (the fixture).  Log-mel
    values of magnitude up to 25: they agree to float32 rounding."""
    import os

    z = np.load(os.path.join(golden_dir, "fbank_hf_speech2text.npz"))
    n = 0
    for k in z.files:
        if not k.startswith("in::wave"):
            continue
        w = z[k].astype(np.float64) * 2.0 ** 15
        ref = z["out::fbank" + k[len("in::wave"):]]
        got = O.kaldi_fbank(w)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)
        n += 1
    assert n == 5


def test_oracle_mel_banks_partition():
    mb = O.kaldi_mel_banks(80, 512, 16000.0)
    assert mb.shape == (80, 257) and (mb >= 0).all() and mb[:, 256].max() == 0.0
    # neighbouring triangles overlap so that interior FFT bins get total weight 1
    tot = mb.sum(0)
    k = np.arange(257) * 16000.0 / 512
    inner = (k > 100) & (k < 7700)  # between the first and the last filter centre (42 Hz .. 7745 Hz)
    np.testing.assert_allclose(tot[inner], 1.0, atol=1e-9)


def test_oracle_cmvn_formula():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(57, 80, generator=g) * 3 + 1.5
    y = O.utterance_cmvn(x)
    xn = x.numpy().astype(np.float64)
    mean = xn.mean(0)
    std = np.sqrt(np.maximum((xn ** 2).sum(0) / xn.shape[0] - mean ** 2, 1e-10))
    np.testing.assert_allclose(y.numpy(), (xn - mean) / std, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_fbank_kernel_matches_oracle():
    from s2t_amd import audio as A
    rng = np.random.default_rng(3)
    lens = [16000, 12345, 400, 399, 8000]
    waves = []
    for i, n in enumerate(lens):
        w = _tone(200.0 * (i + 1), n) + rng.normal(0, 300.0, n) + 40.0 * i
        waves.append(torch.from_numpy(w.astype(np.float32)).cuda())
    feat, n_frames = A.fbank_batch(waves, sample_rate=16000, n_bins=80)
    torch.cuda.synchronize()
    assert feat.shape == (5, 98, 80)
    assert n_frames.tolist() == [98, 75, 1, 0, 48]
    for i, n in enumerate(lens):
        ref = O.kaldi_fbank(waves[i].cpu().numpy().astype(np.float64))
        got = feat[i].cpu().numpy()
        T = ref.shape[0]
        # fp32 FFT / mel accumulation vs float64: log-mel values of O(10), agreement to ~1e-4 absolute
        np.testing.assert_allclose(got[:T], ref, rtol=0, atol=2e-3)
        assert np.abs(got[T:]).max(initial=0.0) == 0.0  # collater padding
    single = A.get_torchaudio_fbank(waves[1].cpu().numpy(), 16000, n_bins=80)
    np.testing.assert_allclose(single, feat[1, :75].cpu().numpy(), atol=1e-6)


@pytest.mark.gpu
def test_fbank_kernel_matches_the_hf_speech2text_fixture(golden_dir):
    """s2t_fbank against the independent restatement's vectors (see the module docstring): fp32 FFT and mel accumulation on the
    GPU against transformers' float32 numpy: 2e-3 absolute on log-mel values (the bound of the oracle comparison above)."""
    import os

    from s2t_amd import audio as A

    z = np.load(os.path.join(golden_dir, "fbank_hf_speech2text.npz"))
    names = sorted(k for k in z.files if k.startswith("in::wave"))
    waves = [torch.from_numpy(z[k] * np.float32(2.0 ** 15)).cuda() for k in names]
    feat, n_frames = A.fbank_batch(waves, sample_rate=16000, n_bins=80)
    torch.cuda.synchronize()
    for i, k in enumerate(names):
        ref = z["out::fbank" + k[len("in::wave"):]]
        assert int(n_frames[i]) == ref.shape[0]
        got = feat[i, :ref.shape[0]].cpu().numpy()
        # fp32 FFT on the GPU against float32 numpy: a low-energy bin (log-mel ~1 where the tones sit at ~20) amplifies the rounding of
        # its few-ulp power sum — measured on MI355X: 1 of 15 840 values at 2.3e-3, everything else below 2e-3, mean 3e-5
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-3)
        assert float(np.abs(got - ref).mean()) < 3e-4


@pytest.mark.gpu
def test_cmvn_kernel_matches_oracle():
    from s2t_amd import audio as A
    g = torch.Generator().manual_seed(1)
    B, T, Cf = 4, 300, 80
    x = torch.randn(B, T, Cf, generator=g) * 4 + 2
    n = torch.tensor([300, 123, 1, 77], dtype=torch.int32)
    for b in range(B):
        x[b, n[b]:] = 0
    for means, vars_ in ((True, True), (True, False), (False, True)):
        tr = A.UtteranceCMVN(means, vars_)
        y = tr.apply_batch(x.cuda(), n.cuda()).cpu()
        for b in range(B):
            ref = O.utterance_cmvn(x[b, :n[b]].double(), means, vars_)
            np.testing.assert_allclose(y[b, :n[b]].numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
            assert y[b, n[b]:].abs().max().item() == 0.0 if n[b] < T else True
    one = A.get_audio_feature_transform("utterance_cmvn").from_config_dict({"norm_vars": True})(x[1, :123].numpy())
    np.testing.assert_allclose(one, O.utterance_cmvn(x[1, :123].double()).numpy(), rtol=1e-4, atol=1e-4)


def _specaug_cases(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "specaugment.npz"))
    for i in range(int(z["cfg::n"])):
        seed, fn, ff, tn, tt, tp, mv = z["cfg::params_%d" % i].tolist()
        yield z["in::x_%d" % i], z["out::y_%d" % i], int(seed), int(fn), int(ff), int(tn), int(tt), tp, (None if mv < 0 else mv)


def test_oracle_specaugment_matches_reference(golden_dir):
    for x, y, seed, fn, ff, tn, tt, tp, mv in _specaug_cases(golden_dir):
        np.random.seed(seed)
        got = O.spec_augment(x, fn, ff, tn, tt, tp, mv)
        np.testing.assert_array_equal(got, y)
        assert (y != x).any()


def test_oracle_row_resize_agrees_with_aten_bilinear():
    """The time warp of SpecAugment resizes two row blocks with cv2.resize(..., INTER_LINEAR) (specaugment.py:96-112); OpenCV is
    absent here and unpinned in the reference, so ``resize_rows_linear`` restates its convention (half-pixel centres, source rows
    clamped at the edges, no anti-aliasing when shrinking).  An independent implementation of the same convention is at hand:
    ATen's bilinear interpolation with align_corners=False.  They agree to the rounding of the fp32 source coordinate (OpenCV and
    the restatement form it in float, ATen in another order): evidence for the convention, not a pin to OpenCV."""
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    for rows, new in ((37, 50), (50, 37), (100, 101), (64, 3), (5, 40), (120, 119), (1, 7)):
        x = rng.standard_normal((rows, 80)).astype(np.float32)
        a = O.resize_rows_linear(x, new)
        b = F.interpolate(torch.from_numpy(x)[None, None], size=(new, 80), mode="bilinear", align_corners=False)[0, 0].numpy()
        assert a.shape == b.shape == (new, 80)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)


def test_oracle_time_warp_properties():
    """The restated INTER_LINEAR row mapping (cv2 is absent: parity unpinned): same size = identity; a ramp stays the same
    ramp away from the clamped ends (half-pixel centres); the warp keeps the frame count and leaves the feature axis alone."""
    x = np.random.RandomState(0).randn(50, 7).astype(np.float32)
    np.testing.assert_array_equal(O.resize_rows_linear(x, 50), x)
    ramp = np.arange(40, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32)
    up = O.resize_rows_linear(ramp, 80)
    np.testing.assert_allclose(up[2:-2, 0], (np.arange(80) + 0.5)[2:-2] * 0.5 - 0.5, rtol=0, atol=1e-5)
    assert up[0, 0] == 0.0 and up[-1, 0] == 39.0  # clamped, not extrapolated
    np.random.seed(5)
    y = O.spec_augment(x, 0, 0, 0, 0, 0.0, 0.0, time_warp_w=8)
    np.random.seed(5)
    w0, w = np.random.randint(8, 50 - 8), np.random.randint(-7, 8)
    assert y.shape == x.shape and w != 0 and (y != x).any()
    np.testing.assert_array_equal(y[:w0 + w], O.resize_rows_linear(x[:w0], w0 + w))
    np.testing.assert_array_equal(y[w0 + w:], O.resize_rows_linear(x[w0:], 50 - w0 - w))
    # T <= 2W: no warp and no draw
    np.random.seed(5)
    np.testing.assert_array_equal(O.spec_augment(x[:16], 0, 0, 0, 0, 0.0, 0.0, time_warp_w=8), x[:16])


@pytest.mark.gpu
def test_time_warp_kernel_matches_oracle():
    from s2t_amd import audio as A
    rs = np.random.RandomState(3)
    for T, W, fn, ff, tn, tt, tp, mv in ((137, 5, 2, 27, 2, 40, 0.5, 0.0), (300, 40, 1, 10, 1, 100, 1.0, None), (9, 5, 1, 5, 1, 3, 1.0, 0.0)):
        x = rs.randn(T, 80).astype(np.float32)
        tr = A.SpecAugmentTransform(W, fn, ff, tn, tt, tp, mv)
        np.random.seed(11)
        got = tr(x)
        np.random.seed(11)
        ref = O.spec_augment(x, fn, ff, tn, tt, tp, mv, time_warp_w=W)
        if mv is None:
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
        else:
            np.testing.assert_array_equal(got, ref)
    # batched, ragged: draws in batch order (warp, then masks, per utterance); padded frames untouched
    x0 = rs.randn(120, 80).astype(np.float32)
    feat = torch.zeros(3, 130, 80)
    lens = [120, 90, 14]
    for b, n in enumerate(lens):
        feat[b, :n] = torch.from_numpy(x0[:n])
    tr = A.SpecAugmentTransform(10, 1, 20, 1, 30, 1.0, 0.0)
    np.random.seed(2)
    out = tr.apply_batch(feat.cuda(), torch.tensor(lens)).cpu()
    np.random.seed(2)
    for b, n in enumerate(lens):
        np.testing.assert_array_equal(out[b, :n].numpy(), O.spec_augment(x0[:n], 1, 20, 1, 30, 1.0, 0.0, time_warp_w=10))
        assert out[b, n:].abs().max() == 0


@pytest.mark.gpu
def test_specaugment_kernel_matches_reference(golden_dir):
    from s2t_amd import audio as A
    for x, y, seed, fn, ff, tn, tt, tp, mv in _specaug_cases(golden_dir):
        tr = A.SpecAugmentTransform(0, fn, ff, tn, tt, tp, mv)
        np.random.seed(seed)
        got = tr(x)
        if mv is None:  # fill value = utterance mean, accumulated on the device
            np.testing.assert_allclose(got, y, rtol=0, atol=1e-5)
        else:
            np.testing.assert_array_equal(got, y)
    # batched: utterances are drawn in batch order, padded frames stay untouched
    x0, y0, seed, fn, ff, tn, tt, tp, mv = next(_specaug_cases(golden_dir))
    T = x0.shape[0]
    feat = torch.zeros(2, T + 9, 80)
    feat[0, :T] = torch.from_numpy(x0)
    feat[1, :T - 5] = torch.from_numpy(x0[:T - 5])
    tr = A.SpecAugmentTransform(0, fn, ff, tn, tt, tp, mv)
    np.random.seed(seed)
    out = tr.apply_batch(feat.cuda(), torch.tensor([T, T - 5])).cpu()
    np.random.seed(seed)
    r0 = O.spec_augment(x0, fn, ff, tn, tt, tp, mv)
    r1 = O.spec_augment(x0[:T - 5], fn, ff, tn, tt, tp, mv)
    np.testing.assert_array_equal(out[0, :T].numpy(), r0)
    np.testing.assert_array_equal(out[1, :T - 5].numpy(), r1)
    assert out[0, T:].abs().max() == 0 and out[1, T - 5:].abs().max() == 0
