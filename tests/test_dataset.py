"""The reference's on-disk format and batch layout: TSV manifest, features in an
uncompressed zip addressed by byte offset, config yaml, dictionary, ``collater`` — against items and batches produced by
the reference's own SpeechToTextDatasetCreator.from_tsv on the same files (oracle/gen_golden.py: dataset_case)."""
import os
import shutil

import numpy as np
import pytest
import torch

from s2t_amd import speech_to_text_dataset as D

TAGS = ["config_utt", "config_global"]


@pytest.fixture(scope="module")
def root(golden_dir, tmp_path_factory):
    """A copy of tests/golden/s2t_dataset with ``audio_root`` pointing at it (the manifest's audio paths are relative)."""
    dst = str(tmp_path_factory.mktemp("s2t_dataset"))
    src = os.path.join(golden_dir, "s2t_dataset")
    for f in os.listdir(src):
        shutil.copy(os.path.join(src, f), dst)
    for tag in TAGS:
        p = os.path.join(dst, tag + ".yaml")
        text = open(p).read().replace("AUDIO_ROOT", dst)
        open(p, "w").write(text)
    return dst


def _dataset(root, tag):
    d = D.Dictionary.load(os.path.join(root, "dict.txt"))
    cfg = D.S2TDataConfig(os.path.join(root, tag + ".yaml"))
    return D.SpeechToTextDatasetCreator.from_tsv(root, cfg, "train", d, None, None, is_train_split=False), d


def test_manifest_zip_and_dictionary(root, golden_dir):
    z = np.load(os.path.join(golden_dir, "s2t_dataset_expected.npz"))
    ds, d = _dataset(root, "config_utt")
    assert len(d) == 4 + 24 and (d.bos(), d.pad(), d.eos(), d.unk()) == (0, 1, 2, 3)
    assert ds.sizes.tolist() == z["config_utt::sizes"].tolist()
    assert ds.ordered_indices().tolist() == z["config_utt::ordered_indices"].tolist()
    # raw features: the zip members (byte offset / length from the manifest) and the plain .npy path
    member = ds.datasets[0]
    assert member.ordered_indices().tolist() == ds.ordered_indices().tolist()[::-1]  # no ties in this fixture
    plain = D.get_features_or_waveform(member.audio_paths[0])
    assert plain.shape == (member.n_frames[0], 80)
    for i in range(1, len(ds)):
        x = D.get_features_or_waveform(member.audio_paths[i])
        assert x.dtype == np.float32 and x.shape == (member.n_frames[i], 80)
    with pytest.raises(FileNotFoundError):
        D.get_features_or_waveform(os.path.join(root, "missing.zip") + ":0:10")
    assert d.encode_line("w1  oov w3").tolist() == [5, 3, 7, 2]  # unknown word -> <unk>, </s> appended


@pytest.mark.parametrize("tag", TAGS)
def test_collater_layout(root, golden_dir, tag):
    """collater on the reference's items reproduces the reference's batch exactly (sorting, padding, eos shift)."""
    z = np.load(os.path.join(golden_dir, "s2t_dataset_expected.npz"))
    ds, d = _dataset(root, tag)
    idx = z[tag + "::batch_idx"].tolist()
    items = [(i, torch.from_numpy(z[tag + "::item_%d_source" % i]), torch.from_numpy(z[tag + "::item_%d_target" % i]), None)
             for i in idx]
    b = ds.collater(items)
    assert b["id"].tolist() == z[tag + "::id"].tolist()
    assert torch.equal(b["net_input"]["src_tokens"], torch.from_numpy(z[tag + "::src_tokens"]))
    assert b["net_input"]["src_lengths"].tolist() == z[tag + "::src_lengths"].tolist()
    assert torch.equal(b["net_input"]["prev_output_tokens"], torch.from_numpy(z[tag + "::prev_output_tokens"]))
    assert torch.equal(b["target"], torch.from_numpy(z[tag + "::target"]))
    assert b["target_lengths"].tolist() == z[tag + "::target_lengths"].tolist()
    assert b["ntokens"] == int(z[tag + "::ntokens"]) and b["nsentences"] == int(z[tag + "::nsentences"])
    assert b["transcript"]["tokens"] is None and ds.collater([]) == {}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_items_through_the_device_transforms(root, golden_dir, tag):
    """__getitem__ end to end: features from disk, utterance / global CMVN on the device, tokenised targets."""
    z = np.load(os.path.join(golden_dir, "s2t_dataset_expected.npz"))
    ds, _ = _dataset(root, tag)
    idx = z[tag + "::batch_idx"].tolist()
    items = [ds[i] for i in idx]
    for i, it in zip(idx, items):
        assert it[0] == i and it[3] is None
        np.testing.assert_allclose(it[1].numpy(), z[tag + "::item_%d_source" % i], rtol=1e-4, atol=2e-5)
        assert it[2].tolist() == z[tag + "::item_%d_target" % i].tolist()
    b = ds.collater(items)
    np.testing.assert_allclose(b["net_input"]["src_tokens"].numpy(), z[tag + "::src_tokens"], rtol=1e-4, atol=2e-5)
    assert torch.equal(b["net_input"]["prev_output_tokens"], torch.from_numpy(z[tag + "::prev_output_tokens"]))


@pytest.mark.gpu
def test_wav_member_is_featurised_on_the_device(tmp_path):
    """A RIFF/WAVE member of an uncompressed zip goes through the HIP fbank kernel (frame count of snip_edges framing)."""
    import struct
    import zipfile
    rate, n = 16000, 16000
    pcm = (np.sin(np.arange(n) * 2 * np.pi * 440 / rate) * 12000).astype("<i2").tobytes()
    wav = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) \
        + b"data" + struct.pack("<I", len(pcm)) + pcm
    zp = str(tmp_path / "audio.zip")
    with zipfile.ZipFile(zp, "w", zipfile.ZIP_STORED) as zf:
        zf.writestr("a.wav", wav)
    with zipfile.ZipFile(zp) as zf:
        i = zf.infolist()[0]
        off = i.header_offset + 30 + len(i.filename)
    feats = D.get_features_or_waveform("%s:%d:%d" % (zp, off, len(wav)))
    assert feats.shape == (1 + (n - 400) // 160, 80) and np.isfinite(feats).all()
    w, r = D.get_features_or_waveform("%s:%d:%d" % (zp, off, len(wav)), need_waveform=True), rate
    assert w.shape == (1, n) and abs(float(np.abs(w).max()) - 12000 / 32768) < 1e-3
