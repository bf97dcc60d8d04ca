"""CPU-only checks of the host side: C-ABI library loads and exports every declared symbol, state_dict
compatibility with the reference, length/mask logic, flat parameter storage, registry names, and that the
product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import s2t_oracle as O
from s2t_amd import _lib
from s2t_amd import s2t_transformer as M
from s2t_amd.criterions import ctc_targets
from s2t_amd.registry import ARCH_MODEL_REGISTRY, CRITERION_REGISTRY, MODEL_REGISTRY


def test_library_exports_every_declared_symbol():
    protos = _lib.header_prototypes()
    assert len(protos) >= 28 and "s2t_gemm" in protos and "s2t_ctc_loss_bwd" in protos
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), "include/s2t_hip.h declares %s but libs2t_hip.so does not export it" % name
    assert _lib.lib().s2t_version() >= 1


def test_gemm_struct_matches_header_field_order():
    import re
    src = open(_lib.HEADER_PATH).read()
    start = src.index("typedef struct s2t_gemm_args {") + len("typedef struct s2t_gemm_args {")
    body = re.sub(r"/\*.*?\*/", "", src[start:src.index("} s2t_gemm_args;")], flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split(",")
        names.append(parts[0].split()[-1])
        names += [p.strip() for p in parts[1:]]
    assert names == [f[0] for f in _lib.GemmArgs._fields_]


def _build(golden_dir, name):
    from s2t_amd import pdss2t_transformer as PDS
    from s2t_amd import s2t_sate as SATE
    from tests.test_model_parity_gpu import args_from_cfg
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = O.cfg_from_golden(z)
    vocab = z["w::decoder.embed_tokens.weight"].shape[0]
    arch = str(cfg.get("arch", ""))
    cls = PDS.PDSS2TTransformerModel if arch.startswith("pdss2t") else SATE.S2TSATEModel if arch.startswith("s2t_sate") \
        else M.S2TTransformerModel
    model = cls.build_model(args_from_cfg(cfg, vocab), M.FakeTask(vocab))
    return model, z


@pytest.mark.parametrize("name", ["transformer_small", "conformer_small", "pds_small", "pds_conformer_small", "sate_small"])
def test_state_dict_keys_and_roundtrip(golden_dir, name):
    """Checkpoint compatibility: same keys and shapes as the reference's state_dict."""
    model, z = _build(golden_dir, name)
    ref = {k[3:]: z[k] for k in z.files if k.startswith("w::")}
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    sd2 = model.state_dict()
    for k, v in ref.items():
        np.testing.assert_array_equal(sd2[k].numpy(), v)
    # tied weights stay tied (s2t_transformer.py:965-971, transformer.py:901-907)
    assert model.decoder.embed_tokens.weight is model.decoder.output_projection.weight
    if name.startswith("sate"):
        assert model.encoder.acoustic_encoder.ctc.ctc_projection.weight is model.decoder.embed_tokens.weight
        assert model.encoder.textual_encoder.embed_tokens.weight is model.decoder.embed_tokens.weight
    else:
        assert model.encoder.ctc.ctc_projection.weight is model.decoder.embed_tokens.weight
    # conv weights are held [Cout][k][Cin] internally
    if not name.startswith(("pds", "sate")):
        w = model.encoder.subsample.layers[0][0].weight
        np.testing.assert_array_equal(w.detach().numpy(), ref["encoder.subsample.layers.0.0.weight"].transpose(0, 2, 1))


def test_flat_parameters_adjacency_and_views(golden_dir):
    from s2t_amd.flat_params import FlatParameters
    from s2t_amd.functional import fused
    model, z = _build(golden_dir, "conformer_small")
    before = {k: v.clone() for k, v in model.state_dict().items()}
    flat = FlatParameters(model, torch.bfloat16)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    att = model.encoder.layers[0].self_attn
    d = att.linear_q.weight.shape[0]
    w = fused([att.linear_q.weight, att.linear_k.weight, att.linear_v.weight], 3 * d, d)
    assert w.dtype == torch.bfloat16
    assert torch.equal(w[d:2 * d].float(), att.linear_k.weight.data.to(torch.bfloat16).float())
    ca = model.decoder.layers[0].encoder_attn
    wkv = fused([ca.k_proj.weight, ca.v_proj.weight], 2 * d, d)
    assert torch.equal(wkv[d:].float(), ca.v_proj.weight.data.to(torch.bfloat16).float())
    # grads are views of ONE buffer; unique parameters only
    n = sum(p.numel() for p in set(model.parameters()))
    assert flat.numel >= n and flat.grad.numel() == flat.numel
    for p in model.parameters():
        assert p.grad.data_ptr() >= flat.grad.data_ptr()
        assert p.grad.data_ptr() < flat.grad.data_ptr() + flat.numel * 4


def test_registry_names():
    import s2t_amd.pdss2t_transformer, s2t_amd.s2t_sate  # noqa: F401,E401
    for n in ("s2t_transformer", "s2t_ctc", "pdss2t_transformer", "s2t_sate"):
        assert n in MODEL_REGISTRY
    for n in ("s2t_transformer", "s2t_transformer_s", "s2t_ctc", "s2t_ctc_s", "pdss2t_transformer_s_8", "s2t_sate"):
        assert n in ARCH_MODEL_REGISTRY
    assert "label_smoothed_cross_entropy_with_ctc" in CRITERION_REGISTRY


def test_no_cpu_fallback(golden_dir):
    model, z = _build(golden_dir, "transformer_small")
    with pytest.raises(RuntimeError, match="GPU only"):
        model.encoder(torch.from_numpy(z["in::src_tokens"]), torch.from_numpy(z["in::src_lengths"]))


def test_ctc_target_packing():
    t = torch.tensor([[5, 6, 7, 2, 1, 1], [9, 2, 1, 1, 1, 1], [4, 4, 4, 4, 4, 2]])
    m, l = ctc_targets(t, 1, 2)
    assert l.tolist() == [3, 1, 5]
    assert m[0, :3].tolist() == [5, 6, 7] and m[1, :1].tolist() == [9] and m[2, :5].tolist() == [4] * 5
    ref = O.ctc_targets(t)
    assert [r.tolist() for r in ref] == [m[i, : l[i]].tolist() for i in range(3)]


def test_out_lengths_match_oracle():
    from s2t_amd.modules import Conv1dSubsampling
    lens = torch.tensor([1, 2, 3, 4, 5, 399, 400, 1000, 2000])
    assert Conv1dSubsampling.get_out_seq_lens_tensor(lens).tolist() == O.subsampled_lengths(lens).tolist()


def test_position_tables_match_oracle():
    from s2t_amd.modules import rel_pos_table, sinusoidal_table
    assert torch.equal(sinusoidal_table(50, 32), O.sinusoidal_table(50, 32))
    assert torch.equal(rel_pos_table(13, 32), O.rel_pos_table(13, 32))


def test_batch_memo_refresh_cascades_and_owners_of_copies_are_distinct():
    """functional.batch_memo on CPU tensors (no launch involved): (1) a memo computed from ANOTHER memo's outputs is refreshed in the
    same sweep when the root's contents change in place (the cascade of ADVICE round 5; the case that needs it — outputs written
    through raw addresses by the one-launch forms, whose version counters never move — exists on the GPU only:
    tests/test_packed_rows_gpu.py::test_an_eager_pass_over_a_batch_overwritten_in_place_sees_the_new_batch); (2) an unpinned entry
    whose outputs change shape is recomputed afresh; (3) a deep copy of an owner gets a serial of its own; (4) unpinning hands an
    entry back when no newer one exists."""
    import copy

    from s2t_amd import functional as Fn

    class Owner:
        pass

    own = Owner()
    root = torch.tensor([3, 5, 7])

    def lens(r):
        return (r * 2,)

    def total(l):
        return (l.sum().reshape(1),)

    k1, k2 = ("t_lens", Fn.memo_owner(own)), ("t_total", Fn.memo_owner(own))
    (l,) = Fn.batch_memo(k1, (root,), lens)
    (t,) = Fn.batch_memo(k2, (l,), total)
    assert int(t) == 30
    root.copy_(torch.tensor([1, 1, 1]))       # the batch overwritten in place
    (l2,) = Fn.batch_memo(k1, (root,), lens)  # first memo of the pass notices the moved version ...
    assert l2 is l and l.tolist() == [2, 2, 2]
    (t2,) = Fn.batch_memo(k2, (l,), total)    # ... and the dependent one was refreshed in the same sweep
    assert t2 is t and int(t) == 6, int(t)
    # (2) a shape change of an unpinned entry
    def ragged(r):
        return (torch.arange(int(r.sum())),)
    k3 = ("t_ragged", Fn.memo_owner(own))
    (a,) = Fn.batch_memo(k3, (root,), ragged)
    assert a.numel() == 3
    root.copy_(torch.tensor([2, 2, 2]))
    (b,) = Fn.batch_memo(k3, (root,), ragged)
    assert b.numel() == 6
    # (3) a copied owner does not share the original's keys
    twin = copy.deepcopy(own)
    assert Fn.memo_owner(twin) != Fn.memo_owner(own) and Fn.memo_owner(own) == k1[1]
    # (4) pin, then unpin: the entry is back in the table (a re-capture over the same batch finds it)
    Fn.pin_batch_memos([root], ("test-owner",))
    assert k1 in Fn._MEMO_PINNED and k1 not in Fn._MEMO
    Fn.unpin_batch_memos(("test-owner",))
    assert k1 in Fn._MEMO and k1 not in Fn._MEMO_PINNED
    (l3,) = Fn.batch_memo(k1, (root,), lens)
    assert l3 is l
    for k in (k1, k2, k3):
        Fn._MEMO.pop(k, None)
