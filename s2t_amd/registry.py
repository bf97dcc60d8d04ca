"""Registration shim with fairseq's decorator names (fairseq/models/__init__.py:101-197,
fairseq/criterions/__init__.py).

Without fairseq (the GPU box receives this repository only) the decorators fill the small local registries the bundled
harness (``s2t_amd.trainer``, ``bench.py``) uses.

With fairseq importable — ``--user-dir /path/to/s2t_amd`` on ``fairseq_cli/train.py`` / ``generate.py``
(fairseq/utils.py:436-467) — the reference has ALREADY registered every name this package provides, and fairseq raises on
duplicates (models/__init__.py:121-141).  Nothing is swallowed; the policy is explicit:

  * default: each class is registered under ``<name>_hip`` and each architecture under ``<arch>_hip``
    (``--arch s2t_transformer_s_hip --criterion label_smoothed_cross_entropy_with_ctc_hip``);
  * ``S2T_AMD_OVERRIDE=1``: the reference's registry entries are REPLACED, so unchanged recipes (``--arch
    s2t_transformer_s``) run the HIP path; a warning names every replaced entry.

Any other registration error propagates.  ``SHADOWED`` / ``REPLACED`` record what happened for the seam test.
"""
import logging
import os

logger = logging.getLogger("s2t_amd.registry")

MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}
CRITERION_REGISTRY = {}
SHADOWED = []   # (kind, reference name, name ours is reachable under)
REPLACED = []   # (kind, name)
SUFFIX = "_hip"
REFERENCE_CLASSES = {}  # model name -> the reference's class (kept when its registry entry is replaced)

try:
    import fairseq.criterions as _fc
    import fairseq.models as _fm

    HAVE_FAIRSEQ = True
except ImportError:
    HAVE_FAIRSEQ = False


def override_requested():
    return os.environ.get("S2T_AMD_OVERRIDE", "0") == "1"


def register_model(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        if not HAVE_FAIRSEQ:
            return cls
        if name not in _fm.MODEL_REGISTRY:
            return _fm.register_model(name)(cls)
        REFERENCE_CLASSES[name] = _fm.MODEL_REGISTRY[name]
        if override_requested():
            _fm.MODEL_REGISTRY[name] = cls
            REPLACED.append(("model", name))
            logger.warning("s2t_amd: model '%s' now resolves to the MI355X implementation %s", name, cls.__name__)
            return cls
        _fm.register_model(name + SUFFIX)(cls)
        SHADOWED.append(("model", name, name + SUFFIX))
        return cls

    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        cls = MODEL_REGISTRY[model_name]
        ARCH_MODEL_REGISTRY[arch_name] = cls
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        if not HAVE_FAIRSEQ:
            return fn
        if arch_name not in _fm.ARCH_MODEL_REGISTRY:
            return _fm.register_model_architecture(model_name, arch_name)(fn)
        if override_requested():
            _fm.ARCH_MODEL_REGISTRY[arch_name] = cls
            _fm.ARCH_CONFIG_REGISTRY[arch_name] = fn
            REPLACED.append(("arch", arch_name))
            return fn
        target = model_name + SUFFIX if (model_name + SUFFIX) in _fm.MODEL_REGISTRY else model_name
        _fm.register_model_architecture(target, arch_name + SUFFIX)(fn)
        SHADOWED.append(("arch", arch_name, arch_name + SUFFIX))
        return fn

    return deco


def register_criterion(name):
    def deco(cls):
        CRITERION_REGISTRY[name] = cls
        if not HAVE_FAIRSEQ:
            return cls
        reg = _fc.CRITERION_REGISTRY
        if name not in reg:
            return _fc.register_criterion(name)(cls)
        if override_requested():
            reg[name] = cls
            REPLACED.append(("criterion", name))
            logger.warning("s2t_amd: criterion '%s' now resolves to the MI355X implementation %s", name, cls.__name__)
            return cls
        # the criterion registry checks the base class and the class NAME for duplicates (fairseq/registry.py:70-90)
        alias = type(cls.__name__ + "Hip", (cls,), {"__doc__": cls.__doc__})
        _fc.register_criterion(name + SUFFIX)(alias)
        SHADOWED.append(("criterion", name, name + SUFFIX))
        return cls

    return deco


def model_base():
    """Base class of the HIP models: fairseq's BaseFairseqModel when fairseq is there (its registry insists,
    models/__init__.py:131-135), plain nn.Module otherwise."""
    if HAVE_FAIRSEQ:
        return _fm.BaseFairseqModel
    import torch.nn as nn

    return nn.Module


def criterion_base():
    """FairseqCriterion when fairseq is there (its registry insists on the base class, fairseq/registry.py:77-80), else an
    nn.Module with the same constructor contract (``__init__(task)`` keeps the task and the target padding index)."""
    if HAVE_FAIRSEQ:
        return _fc.FairseqCriterion
    import torch.nn as nn

    class _Criterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task
            if hasattr(task, "target_dictionary"):
                tgt = task.target_dictionary
                self.padding_idx = tgt.pad() if tgt is not None else -100

    return _Criterion


def reference_model_class(name):
    """The reference's class registered under ``name`` (None without fairseq): its ``add_args`` defines the command-line
    flags of the recipes, which the HIP classes accept by delegation."""
    if not HAVE_FAIRSEQ:
        return None
    cls = REFERENCE_CLASSES.get(name) or _fm.MODEL_REGISTRY.get(name)
    return None if cls is None or cls in MODEL_REGISTRY.values() else cls
