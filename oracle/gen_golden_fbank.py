#!/usr/bin/env python3
"""Fixture for row a1 (log-mel filterbank): tests/golden/fbank_hf_speech2text.npz.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  The reference computes its features with
``torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins=80, sample_frequency=sr)`` (fairseq/data/audio/audio_utils.py:59-79);
torchaudio is a third-party dependency that is neither vendored nor pinned in the reference (setup.py:206) and is not installed in
this image, and the reference's tests hold no fbank vector — so the oracle's ``kaldi_fbank`` cannot be pinned against the reference
itself.  What IS installed is Hugging Face transformers, whose ``Speech2TextFeatureExtractor`` is the port of exactly this fairseq S2T
front-end and carries a numpy implementation of the same Kaldi call for machines without torchaudio
(transformers/models/speech_to_text/feature_extraction_speech_to_text.py: povey window, 25 ms / 10 ms frames, remove_dc_offset,
pre-emphasis 0.97, 512-point power spectrum, 80 Kaldi-scale mel banks from 20 Hz, log with the float32-epsilon floor).  This script
runs THAT implementation (an independent third party's restatement, not this repo's) on seeded waveforms and stores inputs and
outputs; tests/test_frontend.py checks the oracle — and on the GPU the HIP kernel — against them.  It does not make a1 "pinned to the
reference": it shows that two independent restatements of the same published algorithm agree to float32 rounding.

usage: python oracle/gen_golden_fbank.py [outdir=tests/golden]      (needs: transformers without torchaudio present)"""
import os
import sys

import numpy as np


def waves():
    rng = np.random.default_rng(20261005)
    out = []
    for i, n in enumerate((400, 8000, 12345, 16000, 31999)):
        t = np.arange(n) / 16000.0
        w = 0.25 * np.sin(2 * np.pi * (180.0 * (i + 1)) * t) + 0.1 * np.sin(2 * np.pi * 2310.0 * t + 0.3) + 0.02 * rng.standard_normal(n) + 0.001 * i
        out.append(np.clip(w, -1.0, 1.0).astype(np.float32))
    return out


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    import transformers
    from transformers.models.speech_to_text.feature_extraction_speech_to_text import Speech2TextFeatureExtractor
    from transformers.utils import is_speech_available

    assert not is_speech_available(), "torchaudio is present: the extractor would call it instead of its numpy port"
    fe = Speech2TextFeatureExtractor(do_ceptral_normalize=False)  # dither 0, 80 bins, 16 kHz: the reference's arguments
    data = {"meta::transformers_version": np.array(transformers.__version__), "meta::sample_rate": np.array(16000)}
    for i, w in enumerate(waves()):
        data["in::wave%d" % i] = w                                  # in [-1, 1]; the reference scales by 2**15 before the call
        data["out::fbank%d" % i] = fe._extract_fbank_features(w.copy()).astype(np.float32)
    np.savez_compressed(os.path.join(outdir, "fbank_hf_speech2text.npz"), **data)
    print("wrote", os.path.join(outdir, "fbank_hf_speech2text.npz"), {k: v.shape for k, v in data.items() if k.startswith("out::")})


if __name__ == "__main__":
    main()
