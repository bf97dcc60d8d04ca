"""Inert stand-ins for third-party packages the reference imports but this image lacks.

TEST INFRASTRUCTURE ONLY (used by oracle/gen_golden.py in the build container to import
/root/reference and dump golden vectors).  Nothing here is shipped or on the product path.
The stand-ins carry no arithmetic: they only let `import fairseq` succeed.
"""
import sys
import types


class _Anything:
    """Attribute sink: any attribute / call / subscript returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __iter__(self):
        return iter(())


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package so that submodule imports resolve
    m.__getattr__ = _missing_attr  # PEP 562: unknown names resolve to an inert sink class
    sys.modules[name] = m
    return m


def _missing_attr(name):
    if name.startswith("__"):
        raise AttributeError(name)
    return _Anything


def install():
    # never let the reference JIT-build (and hipify in place) its CUDA extension
    sys.modules["fairseq.torch_imputer"] = None

    om = _mod("omegaconf")
    om.II = lambda x: x
    om.MISSING = "???"

    class DictConfig(dict):
        pass

    class _OmegaConf:
        @staticmethod
        def is_config(x):
            return False

        @staticmethod
        def create(x=None):
            return x

        @staticmethod
        def set_struct(*a, **k):
            return None

        @staticmethod
        def to_container(x, **k):
            return x

        @staticmethod
        def merge(*a):
            return a[0]

    class open_dict:
        def __init__(self, *a):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    om.DictConfig = DictConfig
    om.OmegaConf = _OmegaConf
    om.open_dict = open_dict
    om._utils = _mod("omegaconf._utils")
    om._utils.is_primitive_type = lambda x: True

    hy = _mod("hydra")
    hc = _mod("hydra.core")
    hcs = _mod("hydra.core.config_store")

    class ConfigStore:
        _inst = None

        @classmethod
        def instance(cls):
            if cls._inst is None:
                cls._inst = cls()
            return cls._inst

        def store(self, *a, **k):
            return None

    hcs.ConfigStore = ConfigStore
    hy.core = hc
    hc.config_store = hcs
    for sub in ("hydra.core.global_hydra", "hydra.experimental", "hydra.core.hydra_config"):
        m = _mod(sub)
        m.GlobalHydra = _Anything
        m.compose = _Anything()
        m.initialize = _Anything()
        m.HydraConfig = _Anything
    hy.initialize = _Anything()
    hy.compose = _Anything()

    ta = _mod("torchaudio")
    tac = _mod("torchaudio.compliance")
    tak = _mod("torchaudio.compliance.kaldi")
    ta.compliance = tac
    tac.kaldi = tak
    for sub in ("torchaudio.sox_effects", "torchaudio.transforms", "torchaudio.functional"):
        _mod(sub)

    ed = _mod("editdistance")
    ed.eval = lambda a, b: 0

    sb = _mod("sacrebleu")
    sb.__version__ = "2.0.0"

    class BLEU:
        TOKENIZERS = ["none", "13a", "intl", "zh", "ja-mecab", "char"]

    sb.BLEU = BLEU
    sbm = _mod("sacrebleu.metrics")
    sbm.BLEU = BLEU
    sbm.CHRF = _Anything
    sb.metrics = sbm

    ca = _mod("configargparse")
    import argparse

    ca.ArgumentParser = argparse.ArgumentParser
    ca.YAMLConfigFileParser = _Anything

    es = _mod("espnet")
    esn = _mod("espnet.nets")
    esc = _mod("espnet.nets.ctc_prefix_score")
    esc.CTCPrefixScore = _Anything
    es.nets = esn
    esn.ctc_prefix_score = esc
