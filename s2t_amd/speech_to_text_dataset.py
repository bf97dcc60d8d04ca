"""Host side of the input pipeline: the reference's on-disk format, items and collated batches
(SURVEY.md §8 rows a2, a3 and f3).

Mirrors fairseq/data/audio/speech_to_text_dataset.py — ``S2TDataConfig`` :30-180, ``get_features_or_waveform`` :193-264
(``<path>`` or ``<zip>:<byte offset>:<byte length>`` into an UNCOMPRESSED zip of ``.npy`` feature matrices or ``.wav``
audio), ``_collate_frames`` :267-285, ``SpeechToTextDataset`` :288-517 (``__getitem__``, ``collater``, ``ordered_indices``),
``SpeechToTextDatasetCreator.from_tsv`` :520-652 (TSV manifest, ``csv.QUOTE_NONE``) — and fairseq/data/dictionary.py
(``Dictionary.load`` / ``encode_line``) and data_utils.collate_tokens (:39-77) as far as the collater needs them.
Same names, arguments and return layout, so a batch from here is interchangeable with one from the reference.

Features stored as ``.npy`` go through the feature transforms on the host in numpy exactly like the reference (they are a
few hundred KB per utterance).  Raw ``.wav`` audio is turned into Kaldi fbank features by the HIP kernel
(``audio.get_torchaudio_fbank`` -> ``s2t_fbank``) and therefore needs the GPU; there is no CPU fbank.
Not built: flac decoding, speed perturbation, temperature-based resampling of concatenated splits.
"""
import csv
import io
import os
import os.path as op
import re
import struct
from copy import deepcopy
from typing import Dict, List

import numpy as np
import torch
import yaml

from . import audio as A


# ------------------------------------------------------------------------------------------------
# dictionary (fairseq/data/dictionary.py)
# ------------------------------------------------------------------------------------------------
class Dictionary:
    """<s>=0 (the CTC blank), <pad>=1, </s>=2, <unk>=3, then the symbols of ``dict.txt`` ("symbol count" per line)."""

    def __init__(self, bos="<s>", pad="<pad>", eos="</s>", unk="<unk>"):
        self.symbols, self.count, self.indices = [], [], {}
        self.bos_index = self.add_symbol(bos)
        self.pad_index = self.add_symbol(pad)
        self.eos_index = self.add_symbol(eos)
        self.unk_index = self.add_symbol(unk)
        self.nspecial = len(self.symbols)

    def add_symbol(self, word, n=1):
        if word in self.indices:
            self.count[self.indices[word]] += n
            return self.indices[word]
        self.indices[word] = len(self.symbols)
        self.symbols.append(word)
        self.count.append(n)
        return self.indices[word]

    @classmethod
    def load(cls, path):
        d = cls()
        with open(path, encoding="utf-8") as f:
            for line in f:
                word, cnt = line.rstrip().rsplit(" ", 1)
                if word in d.indices:
                    raise RuntimeError("Duplicate word found when loading Dictionary: '%s'" % word)
                d.add_symbol(word, int(cnt))
        return d

    def __len__(self):
        return len(self.symbols)

    def __contains__(self, sym):
        return sym in self.indices

    def __getitem__(self, idx):
        return self.symbols[idx] if idx < len(self.symbols) else self.symbols[self.unk_index]

    def index(self, sym):
        return self.indices.get(sym, self.unk_index)

    def bos(self):
        return self.bos_index

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def encode_line(self, line, add_if_not_exist=False, append_eos=True):
        """dictionary.py:307-337 with the default ``tokenize_line`` (collapse whitespace, split on blanks)."""
        words = re.sub(r"\s+", " ", line).strip().split()
        ids = torch.IntTensor(len(words) + (1 if append_eos else 0))
        for i, w in enumerate(words):
            ids[i] = self.add_symbol(w) if add_if_not_exist else self.index(w)
        if append_eos:
            ids[len(words)] = self.eos_index
        return ids


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False):
    """data/data_utils.py:39-77 (without pad_to_length / pad_to_multiple)."""
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        dst = res[i][size - len(v):] if left_pad else res[i][: len(v)]
        if move_eos_to_beginning:
            dst[0] = v[-1] if eos_idx is None else eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res


# ------------------------------------------------------------------------------------------------
# config (speech_to_text_dataset.py:30-180)
# ------------------------------------------------------------------------------------------------
class S2TDataConfig:
    def __init__(self, yaml_path):
        self.config = {}
        self.yaml_dir = op.dirname(op.abspath(yaml_path))
        if op.isfile(yaml_path):
            with open(yaml_path) as f:
                self.config = yaml.load(f, Loader=yaml.FullLoader) or {}

    def _get(self, k, default=None):
        return self.config.get(k, default)

    vocab_filename = property(lambda s: s._get("vocab_filename"))
    asr_vocab_filename = property(lambda s: s._get("asr_vocab_filename"))
    share_src_and_tgt = property(lambda s: s._get("share_src_and_tgt", False))
    shuffle = property(lambda s: s._get("shuffle", False))
    prepend_tgt_lang_tag = property(lambda s: s._get("prepend_tgt_lang_tag", False))
    input_feat_per_channel = property(lambda s: s._get("input_feat_per_channel", 80))
    input_channels = property(lambda s: s._get("input_channels", 1))
    use_audio_input = property(lambda s: s._get("use_audio_input", False))
    speed_perturb = property(lambda s: s._get("speed_perturb", False))
    audio_root = property(lambda s: s._get("audio_root", ""))

    # -- feature-transform selection (behaviour of speech_to_text_dataset.py:134-180) ------------------------------------
    def _transform_names(self, split, is_train):
        """The transform list of ``split``: its own entry, else the ``_train`` / ``_eval`` wildcard of its kind, else ``*``."""
        table = self.config.get("transforms") or {}
        for key in (split, "_train" if is_train else "_eval", "*"):
            if table.get(key) is not None:
                return list(table[key])
        return []

    def _global_cmvn_stats(self):
        """Where the global CMVN statistics live: ``cmvn_path`` (absolute, or relative to the YAML's directory) wins over a
        ``global_cmvn`` section / the top-level ``stats_npz_path`` shorthand."""
        path = self.config.get("cmvn_path")
        if path:
            return path if op.exists(path) else op.join(self.yaml_dir, path)
        if self.config.get("stats_npz_path"):
            return self.config["stats_npz_path"]
        return (self.config.get("global_cmvn") or {}).get("stats_npz_path")

    def get_feature_transforms(self, split, is_train):
        """Config dictionary for ``CompositeAudioFeatureTransform.from_config_dict`` of one split: the selected transform
        names under ``transforms`` plus each transform's own section.  ``no_specaugment`` drops SpecAugment from training
        splits; ``cmvn: utterance | global`` replaces whichever CMVN the list names by the requested kind (global needs
        its statistics file to exist); ``overwrite_specaug`` overrides individual SpecAugment settings."""
        names = self._transform_names(split, is_train)
        out = deepcopy(self.config)
        if self.config.get("stats_npz_path"):  # shorthand, also recorded on the live config like the reference does
            self.config["global_cmvn"] = out["global_cmvn"] = {"stats_npz_path": self.config["stats_npz_path"]}
        if is_train and self.config.get("no_specaugment", False):
            names = [n for n in names if n != "specaugment"]
        kind = self.config.get("cmvn")
        if kind and any(n.endswith("_cmvn") and n in ("utterance_cmvn", "global_cmvn") for n in names):
            names = [n for n in names if n not in ("utterance_cmvn", "global_cmvn")]
            if kind in ("utterance", "global"):
                names.append(kind + "_cmvn")
            if kind == "global":
                stats = self._global_cmvn_stats()
                if self.config.get("cmvn_path"):
                    out["global_cmvn"] = {"stats_npz_path": stats}
                if not (stats and "global_cmvn" in out and op.exists(out["global_cmvn"]["stats_npz_path"])):
                    raise AssertionError("cmvn: global needs an existing statistics file (cmvn_path / global_cmvn.stats_npz_path)")
        if "utterance_cmvn" in names:
            out["utterance_cmvn"] = dict(out.get("utterance_cmvn") or {})
        over = self.config.get("overwrite_specaug")
        if over is not None and "specaugment" in names:
            out["specaugment"].update({k: v for k, v in over.items() if v is not None})
        out["transforms"] = names
        return out


# ------------------------------------------------------------------------------------------------
# features (speech_to_text_dataset.py:183-264)
# ------------------------------------------------------------------------------------------------
_NPY_MAGIC = b"\x93N"      # numpy.lib.format magic b"\x93NUMPY" (the reference tests its first two bytes)
_AUDIO_MAGICS = (b"fL", b"RI")  # "fLaC" / "RIFF"


def is_npy_data(data: bytes) -> bool:
    """Does a zip member hold a ``.npy`` feature matrix?"""
    return bytes(data[:2]) == _NPY_MAGIC


def is_flac_or_wav_data(data: bytes) -> bool:
    """Does a zip member hold raw audio (flac or wav container)?"""
    return bytes(data[:2]) in _AUDIO_MAGICS


def read_from_uncompressed_zip(file_path, offset, file_size) -> bytes:
    """``file_size`` bytes at ``offset`` of a STORED (uncompressed) zip: the member's payload, addressed the way the
    manifests do (``<zip>:<offset>:<length>``)."""
    fd = os.open(file_path, os.O_RDONLY)
    try:
        return os.pread(fd, file_size, offset)
    finally:
        os.close(fd)


def _parse_wav(data: bytes):
    """RIFF/WAVE PCM (16-bit) or IEEE float (32-bit) -> (float32 [channels, samples] in [-1, 1], sample_rate)."""
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("only RIFF/WAVE audio is decoded on this path (flac needs an external decoder)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("malformed wav")
    tag, ch, rate, _, _, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    else:
        raise ValueError("unsupported wav encoding (format %d, %d bits)" % (tag, bits))
    return x.reshape(-1, ch).T.copy(), rate


def get_waveform(path_or_bytes, normalization=True):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    wav, rate = _parse_wav(bytes(data))
    if not normalization:
        wav = wav * 2 ** 15  # audio_utils.py:31-32: int16 range
    return wav, rate


def get_fbank(path_or_bytes, n_bins=80):
    """data/audio/audio_utils.py:82-99: mono, int16-scaled waveform -> Kaldi fbank (here: the HIP kernel)."""
    wav, rate = get_waveform(path_or_bytes, normalization=False)
    wav = wav.mean(axis=0, keepdims=True) if wav.shape[0] > 1 else wav
    feats = A.get_torchaudio_fbank(wav, rate, n_bins)
    return feats.cpu().numpy() if torch.is_tensor(feats) else feats


def get_features_or_waveform(path: str, need_waveform=False):
    """:229-264 — ``<.npy/.wav path>`` or ``<zip path>:<byte offset>:<byte length>``."""
    _path, *extra = path.split(":")
    if not op.exists(_path):
        raise FileNotFoundError("File not found: %s" % _path)
    if len(extra) == 0:
        if need_waveform:
            return get_waveform(_path)
        ext = op.splitext(op.basename(_path))[1]
        if ext not in {".npy", ".flac", ".wav"}:
            raise ValueError('Unsupported file format for "%s"' % _path)
        return np.load(_path) if ext == ".npy" else get_fbank(_path)
    if len(extra) == 2:
        off, size = int(extra[0]), int(extra[1])
        data = read_from_uncompressed_zip(_path, off, size)
        if is_npy_data(data):
            return np.load(io.BytesIO(data))
        if is_flac_or_wav_data(data):
            return get_waveform(data)[0] if need_waveform else get_fbank(data)
        raise ValueError('Unknown file format for "%s"' % _path)
    raise ValueError("Invalid path: %s" % path)


def _collate_frames(frames: List[torch.Tensor], is_audio_input: bool = False) -> torch.Tensor:
    """:267-285 — zero-padded (B, max_len, feat) (or (B, max_len) for raw audio)."""
    max_len = max(f.size(0) for f in frames)
    out = frames[0].new_zeros((len(frames), max_len) if is_audio_input else (len(frames), max_len, frames[0].size(1)))
    for i, v in enumerate(frames):
        out[i, : v.size(0)] = v
    return out


# ------------------------------------------------------------------------------------------------
# dataset (speech_to_text_dataset.py:288-517)
# ------------------------------------------------------------------------------------------------
class SpeechToTextDataset(torch.utils.data.Dataset):
    LANG_TAG_TEMPLATE = "<lang:{}>"

    def __init__(self, split, is_train_split, data_cfg, audio_paths, n_frames, src_texts=None, tgt_texts=None,
                 speakers=None, src_langs=None, tgt_langs=None, ids=None, src_dict=None, tgt_dict=None, pre_tokenizer=None,
                 bpe_tokenizer=None, src_bpe_tokenizer=None):
        self.split, self.is_train_split, self.data_cfg = split, is_train_split, data_cfg
        if data_cfg.speed_perturb:
            raise NotImplementedError("speed perturbation")
        self.audio_paths, self.n_frames = audio_paths, n_frames
        self.n_samples = len(audio_paths)
        if data_cfg.share_src_and_tgt:
            src_texts = tgt_texts
        assert len(n_frames) == self.n_samples > 0
        assert (tgt_dict is None and tgt_texts is None) or (tgt_dict is not None and tgt_texts is not None)
        self.src_texts, self.tgt_texts = src_texts, tgt_texts
        self.src_langs, self.tgt_langs, self.speakers, self.ids = src_langs, tgt_langs, speakers, ids
        self.src_dict, self.tgt_dict = src_dict, tgt_dict
        if data_cfg.prepend_tgt_lang_tag:
            assert tgt_langs is not None and tgt_dict is not None
            assert all(self.LANG_TAG_TEMPLATE.format(t) in tgt_dict for t in set(tgt_langs))
        self.shuffle = data_cfg.shuffle if is_train_split else False
        self.feature_transforms = A.CompositeAudioFeatureTransform.from_config_dict(
            data_cfg.get_feature_transforms(split, is_train_split))
        self.pre_tokenizer, self.bpe_tokenizer, self.src_bpe_tokenizer = pre_tokenizer, bpe_tokenizer, src_bpe_tokenizer

    def tokenize_text(self, text, is_src=False):
        if self.pre_tokenizer is not None:
            text = self.pre_tokenizer.encode(text)
        if self.bpe_tokenizer is not None:
            text = self.src_bpe_tokenizer.encode(text) if is_src else self.bpe_tokenizer.encode(text)
        return text

    def __getitem__(self, index):
        source = get_features_or_waveform(self.audio_paths[index], need_waveform=self.data_cfg.use_audio_input)
        if self.feature_transforms is not None:
            assert not self.data_cfg.use_audio_input
            source = self.feature_transforms(source)
        source = torch.from_numpy(np.asarray(source)).float()
        target = None
        if self.tgt_texts is not None:
            target = self.tgt_dict.encode_line(self.tokenize_text(self.tgt_texts[index]), add_if_not_exist=False,
                                               append_eos=True).long()
            if self.data_cfg.prepend_tgt_lang_tag:
                tag = self.tgt_dict.index(self.LANG_TAG_TEMPLATE.format(self.tgt_langs[index]))
                target = torch.cat((torch.LongTensor([tag]), target), 0)
        transcript = None
        if self.src_dict is not None and self.src_texts is not None and self.src_bpe_tokenizer is not None:
            transcript = self.src_dict.encode_line(self.tokenize_text(self.src_texts[index], True), add_if_not_exist=False,
                                                   append_eos=True).long()
        return index, source, target, transcript

    def __len__(self):
        return self.n_samples

    def collater(self, samples) -> Dict:
        """:411-485 — sorted by descending frame count, zero-padded frames, ``prev_output_tokens`` = target with </s>
        moved to the front."""
        if len(samples) == 0:
            return {}
        indices = torch.tensor([i for i, _, _, _ in samples], dtype=torch.long)
        frames = _collate_frames([s for _, s, _, _ in samples], self.data_cfg.use_audio_input)
        n_frames = torch.tensor([s.size(0) for _, s, _, _ in samples], dtype=torch.long)
        n_frames, order = n_frames.sort(descending=True)
        indices, frames = indices.index_select(0, order), frames.index_select(0, order)
        target = target_lengths = prev_output_tokens = ntokens = None
        if self.tgt_texts is not None:
            tl = [t for _, _, t, _ in samples]
            target = collate_tokens(tl, self.tgt_dict.pad(), self.tgt_dict.eos()).index_select(0, order)
            target_lengths = torch.tensor([t.size(0) for t in tl], dtype=torch.long).index_select(0, order)
            prev_output_tokens = collate_tokens(tl, self.tgt_dict.pad(), self.tgt_dict.eos(),
                                                move_eos_to_beginning=True).index_select(0, order)
            ntokens = sum(t.size(0) for t in tl)
        transcript = transcript_lengths = transcript_ntokens = None
        if self.src_dict is not None and self.src_texts is not None:
            sl = [t for _, _, _, t in samples]
            transcript = collate_tokens(sl, self.src_dict.pad(), self.src_dict.eos()).index_select(0, order)
            transcript_lengths = torch.tensor([t.size(0) for t in sl], dtype=torch.long).index_select(0, order)
            transcript_ntokens = sum(t.size(0) for t in sl)
        return {"id": indices,
                "net_input": {"src_tokens": frames, "src_lengths": n_frames, "prev_output_tokens": prev_output_tokens},
                "transcript": {"tokens": transcript, "lengths": transcript_lengths, "ntokens": transcript_ntokens},
                "target": target, "target_lengths": target_lengths, "ntokens": ntokens, "nsentences": len(samples)}

    def num_tokens(self, index):
        return self.n_frames[index]

    def size(self, index):
        t_len = len(self.tokenize_text(self.tgt_texts[index]).split(" ")) if self.tgt_texts is not None else 0
        return self.n_frames[index], t_len

    @property
    def sizes(self):
        return np.array(self.n_frames)

    def ordered_indices(self):
        """:504-512 — descending frame count, ties in original (or shuffled) order."""
        order = [np.random.permutation(len(self))] if self.shuffle else [np.arange(len(self))]
        order.append([-n for n in self.n_frames])
        return np.lexsort(order)


class SpeechToTextDatasetCreator:
    KEY_ID, KEY_AUDIO, KEY_N_FRAMES, KEY_TGT_TEXT = "id", "audio", "n_frames", "tgt_text"
    KEY_SPEAKER, KEY_SRC_TEXT, KEY_SRC_LANG, KEY_TGT_LANG = "speaker", "src_text", "src_lang", "tgt_lang"
    DEFAULT_SPEAKER = DEFAULT_SRC_TEXT = DEFAULT_LANG = ""

    @classmethod
    def _from_list(cls, split_name, is_train_split, samples, data_cfg, tgt_dict, pre_tokenizer, bpe_tokenizer, src_dict=None,
                   src_bpe_tokenizer=None):
        g = lambda key, default=None: [ss.get(key, default) if default is not None else ss[key] for ss in samples]  # noqa: E731
        return SpeechToTextDataset(
            split_name, is_train_split, data_cfg, [op.join(data_cfg.audio_root, a) for a in g(cls.KEY_AUDIO)],
            [int(n) for n in g(cls.KEY_N_FRAMES)], [ss.get(cls.KEY_SRC_TEXT, cls.DEFAULT_SRC_TEXT) for ss in samples],
            g(cls.KEY_TGT_TEXT), [ss.get(cls.KEY_SPEAKER, cls.DEFAULT_SPEAKER) for ss in samples],
            [ss.get(cls.KEY_SRC_LANG, cls.DEFAULT_LANG) for ss in samples],
            [ss.get(cls.KEY_TGT_LANG, cls.DEFAULT_LANG) for ss in samples], g(cls.KEY_ID), src_dict, tgt_dict, pre_tokenizer,
            bpe_tokenizer, src_bpe_tokenizer)

    @classmethod
    def from_tsv(cls, root, data_cfg, splits, tgt_dict, pre_tokenizer, bpe_tokenizer, is_train_split, epoch=1, seed=1,
                 src_dict=None, src_bpe_tokenizer=None):
        """:594-652 — one TSV per split, tab separated, no quoting; several splits are concatenated."""
        datasets = []
        for split in splits.split(","):
            tsv_path = op.join(root, "%s.tsv" % split)
            if not op.isfile(tsv_path):
                raise FileNotFoundError("Dataset not found: %s" % tsv_path)
            with open(tsv_path) as f:
                reader = csv.DictReader(f, delimiter="\t", quotechar=None, doublequote=False, lineterminator="\n",
                                        quoting=csv.QUOTE_NONE)
                samples = [dict(e) for e in reader]
            datasets.append(cls._from_list(split, is_train_split, samples, data_cfg, tgt_dict, pre_tokenizer, bpe_tokenizer,
                                           src_dict, src_bpe_tokenizer))
        if is_train_split and len(datasets) > 1 and data_cfg._get("sampling_alpha", 1.0) != 1.0:
            raise NotImplementedError("temperature-based resampling of concatenated splits")
        return ConcatDataset(datasets)  # also for a single split, like the reference (:652)


class ConcatDataset(torch.utils.data.ConcatDataset):
    """fairseq/data/concat_dataset.py as far as the speech task uses it: items and ``collater`` of the member datasets;
    ``ordered_indices`` = ASCENDING frame count (``np.argsort(sizes)``, :87-106) — note that this, not the member's own
    descending order, is what the reference's batch sampler sees."""

    @property
    def sizes(self):
        return np.concatenate([np.asarray(ds.sizes) for ds in self.datasets])

    def ordered_indices(self):
        return np.argsort(self.sizes)

    def collater(self, samples, **extra):
        return self.datasets[0].collater(samples, **extra)

    def num_tokens(self, index):
        return int(self.sizes[index])

    def size(self, index):
        i = int(np.searchsorted(self.cumulative_sizes, index, side="right"))
        return self.datasets[i].size(index - (self.cumulative_sizes[i - 1] if i else 0))
