"""Registration shim with fairseq's decorator names (fairseq/models/__init__.py:101-197,
fairseq/criterions/__init__.py).

When fairseq is importable the decorators delegate to it, so the classes below are discovered through
``--user-dir s2t_amd`` by ``fairseq_cli/train.py`` / ``generate.py`` unchanged.  When it is
not (the GPU box receives this repository only) they fill the small local registries that the bundled
harness (``s2t_amd.trainer``, ``bench.py``) uses.
"""
MODEL_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}
CRITERION_REGISTRY = {}

try:  # pragma: no cover - exercised only where fairseq is installed
    from fairseq.models import register_model as _fs_register_model
    from fairseq.models import register_model_architecture as _fs_register_arch
    from fairseq.criterions import register_criterion as _fs_register_criterion

    HAVE_FAIRSEQ = True
except Exception:  # noqa: BLE001
    HAVE_FAIRSEQ = False


def register_model(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        if HAVE_FAIRSEQ:  # pragma: no cover
            try:
                return _fs_register_model(name)(cls)
            except ValueError:
                return cls  # the reference already registered this name; keep ours available locally
        return cls

    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        if HAVE_FAIRSEQ:  # pragma: no cover
            try:
                return _fs_register_arch(model_name, arch_name)(fn)
            except ValueError:
                return fn
        return fn

    return deco


def register_criterion(name):
    def deco(cls):
        CRITERION_REGISTRY[name] = cls
        if HAVE_FAIRSEQ:  # pragma: no cover
            try:
                return _fs_register_criterion(name)(cls)
            except ValueError:
                return cls
        return cls

    return deco
