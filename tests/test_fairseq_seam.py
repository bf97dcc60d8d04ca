"""The fairseq seam, checked against the REFERENCE ITSELF in the build container: with the reference
importable (third-party stand-ins of oracle/ref_stubs.py; the reference never travels, so the test skips where
/root/reference is absent), importing this package the way ``--user-dir`` does (fairseq/utils.py:436-467)

  * registers every HIP class under a deterministic name next to the reference's (``<name>_hip``), or, with
    S2T_AMD_OVERRIDE=1, in place of it — never silently shadowed (fairseq/models/__init__.py:121-141 raises on duplicates);
  * accepts the model flags of the recipe YAMLs (egs/mustc/asr/conf/{base,ctc,conformer}.yaml) through the delegated
    ``add_args``;
  * builds from the parsed namespace and loads a state dict saved by the reference model STRICTLY.

Runs in a child process: the stand-ins must not leak into the test process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CHILD = r'''
import argparse, os, sys
import ref_stubs
ref_stubs.install()
import torch, yaml
import fairseq
from fairseq import models as fm, criterions as fc
from fairseq.data import Dictionary
ref_model = fm.MODEL_REGISTRY["s2t_transformer"]
ref_crit = fc.CRITERION_REGISTRY["label_smoothed_cross_entropy_with_ctc"]

# --user-dir semantics: fairseq.utils.import_user_module imports the package by path
import importlib
sys.path.insert(0, os.environ["S2T_REPO"])
for m in ("s2t_amd.s2t_transformer", "s2t_amd.pdss2t_transformer", "s2t_amd.s2t_sate", "s2t_amd.criterions"):
    importlib.import_module(m)
from s2t_amd import registry as R, s2t_transformer as M

override = os.environ.get("S2T_AMD_OVERRIDE") == "1"
names = ["s2t_transformer", "s2t_ctc", "pdss2t_transformer", "s2t_sate"]
if override:
    for n in names:
        assert fm.MODEL_REGISTRY[n] is R.MODEL_REGISTRY[n], n
    assert fm.ARCH_MODEL_REGISTRY["s2t_transformer_s"] is M.S2TTransformerModel
    assert fc.CRITERION_REGISTRY["label_smoothed_cross_entropy_with_ctc"].__module__.startswith("s2t_amd")
    assert ("model", "s2t_transformer") in R.REPLACED and not R.SHADOWED
    arch = "s2t_transformer_s"
else:
    assert fm.MODEL_REGISTRY["s2t_transformer"] is ref_model            # the reference's entries are untouched ...
    for n in names:
        assert fm.MODEL_REGISTRY[n + "_hip"] is R.MODEL_REGISTRY[n], n   # ... ours sit beside them under <name>_hip
    assert fm.ARCH_MODEL_REGISTRY["s2t_transformer_s_hip"] is M.S2TTransformerModel
    assert fm.ARCH_MODEL_REGISTRY["pdss2t_transformer_s_8_hip"].__module__ == "s2t_amd.pdss2t_transformer"
    assert fc.CRITERION_REGISTRY["label_smoothed_cross_entropy_with_ctc"] is ref_crit
    assert issubclass(fc.CRITERION_REGISTRY["label_smoothed_cross_entropy_with_ctc_hip"], R.CRITERION_REGISTRY["label_smoothed_cross_entropy_with_ctc"])
    assert ("model", "s2t_transformer", "s2t_transformer_hip") in R.SHADOWED and not R.REPLACED
    arch = "s2t_transformer_s_hip"
cls = fm.ARCH_MODEL_REGISTRY[arch]
assert issubclass(cls, fm.BaseFairseqModel)

# the model flags of the recipe YAMLs parse through OUR class's add_args (delegated to the reference's definitions)
# like options.parse_args_and_arch (fairseq/options.py:179-189): model flags live in a group whose defaults are SUPPRESSED,
# so that whatever the command line leaves out is filled in by the architecture function
parser = argparse.ArgumentParser(allow_abbrev=False, argument_default=argparse.SUPPRESS)
group = parser.add_argument_group("Model-specific configuration", argument_default=argparse.SUPPRESS)
cls.add_args(group)
known = {a.dest for a in parser._actions}
conf = {}
for f in ("base.yaml", "ctc.yaml", "conformer.yaml"):
    with open(os.path.join(os.environ["S2T_REF"], "egs/mustc/asr/conf", f)) as fh:
        for line in fh.read().replace("Truectc", "True\nctc").replace("Truearch", "True\narch").splitlines():
            line = line.split("#")[0].strip()
            if line and ":" in line:
                k, v = line.split(":", 1)
                conf[k.strip()] = v.strip()
argv = []
model_keys = []
for k, v in conf.items():
    dest = k.replace("-", "_")
    if dest in known:
        model_keys.append(k)
        act = next(a for a in parser._actions if a.dest == dest)
        if act.nargs == 0:
            if v == "True":
                argv.append("--" + k)
        else:
            argv += ["--" + k, v]
for must in ("macaron-style", "use-cnn-module", "cnn-module-kernel", "encoder-attention-type", "encoder-activation-fn",
             "layer-padding-mask", "share-ctc-and-embed", "encoder-layers", "subsampling-filter", "dropout"):
    assert must in model_keys, must
args = parser.parse_args(argv)
args.arch = arch
args.ctc_weight = float(conf["ctc-weight"])  # a criterion flag that the model reads too (s2t_transformer.py:951-963)
args.input_feat_per_channel, args.input_channels = 80, 1
args.max_source_positions, args.max_target_positions = 6000, 1024
args.encoder_layers, args.decoder_layers = 2, 2   # a quick build; every other value is the recipe's
fm.ARCH_CONFIG_REGISTRY[arch](args)
assert args.macaron_style and args.use_cnn_module and args.encoder_attention_type == "rel_pos"

d = Dictionary()
for i in range(36):
    d.add_symbol("w%d" % i)

class Task:
    source_dictionary = target_dictionary = src_dict = tgt_dict = d
    def get_source_dictionary(self, i):
        return d

import copy
ref_args = copy.deepcopy(args)
ref_args.arch = "s2t_transformer_s"
torch.manual_seed(0)
ref = ref_model.build_model(ref_args, Task())
ours = cls.build_model(args, Task())
sd = ref.state_dict()
missing, unexpected = ours.load_state_dict(sd, strict=True)
assert not missing and not unexpected
assert set(ours.state_dict().keys()) == set(sd.keys())
for k, v in ours.state_dict().items():
    assert v.shape == sd[k].shape, k
# the reference trainer's model.bfloat16() must not cast the fp32 masters (the bf16 shadow is made by prepare())
ours.bfloat16()
assert next(ours.parameters()).dtype == torch.float32 and ours._compute_dtype == torch.bfloat16
# ... and a forward on the CPU fails loudly instead of silently running something else
try:
    ours(torch.zeros(1, 20, 80), torch.tensor([20]), torch.tensor([[2, 5]]))
    raise SystemExit("forward on the CPU did not raise")
except RuntimeError as e:
    assert "GPU only" in str(e), e
print("SEAM_OK", arch, len(model_keys))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fairseq")), reason="the reference is only present in the build container")
@pytest.mark.parametrize("override", ["0", "1"])
def test_user_dir_registration_flags_and_checkpoint(override, tmp_path):
    env = dict(os.environ)
    env.update({"PYTHONPATH": REF + os.pathsep + os.path.join(ROOT, "oracle"), "PYTHONDONTWRITEBYTECODE": "1",
                "S2T_REPO": ROOT, "S2T_REF": REF, "S2T_AMD_OVERRIDE": override})
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SEAM_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "fairseq")), reason="the reference is only present in the build container")
@pytest.mark.parametrize("group", ["nast", "interctc"])
def test_committed_fixtures_regenerate_from_the_reference(group, tmp_path):
    """``oracle/gen_golden.py --check``: the committed golden vectors of a fixture group are what the reference produces
    TODAY (regenerated into a temporary directory from the imported reference and compared key by key).  The ``nast`` group
    holds the fixture whose training pass draws from numpy's global generator (seeded since round 3)."""
    env = dict(os.environ)
    env.update({"PYTHONPATH": REF + os.pathsep + os.path.join(ROOT, "oracle"), "PYTHONDONTWRITEBYTECODE": "1", "GOLDEN_ONLY": group})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--check", os.path.join(ROOT, "tests", "golden")],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GOLDEN_CHECK OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
