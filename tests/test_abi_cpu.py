"""CPU suite: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mgvae.h declares; the product path refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return g


def test_header_symbols_are_exported_and_bound(built):
    from hipops import _native as nat
    hdr = open(os.path.join(ROOT, "include", "mgvae.h")).read()
    declared = set(re.findall(r"\b(mgvae_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert nat.lib().mgvae_strerror(-1).decode().startswith("invalid")
    assert nat.lib().mgvae_bce_partial_floats() > 0
    assert nat.lib().mgvae_cbam_save_floats(2, 32, 4, 5) == 4 * 2 * 32 + 2 * 2 * 2 + 4 * 2 * 20


def test_conv_desc_layout_matches_header(built):
    from hipops import _native as nat
    assert ctypes.sizeof(nat.ConvDesc) == 19 * 4
    assert ctypes.sizeof(nat.ProfRec) == 32


def test_no_cpu_fallback(built):
    from graph.encoder import Encoder
    enc = Encoder([64, 128, 256, 512, 1024])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(1, 1, 96, 60))


def test_invalid_descriptor_is_rejected_without_a_gpu(built):
    from hipops import _native as nat
    d = nat.ConvDesc(1, 4, 8, 8, 4, 9, 8, 3, 3, 1, 1, 1, 1, 4, 0, 4, 0, 0, 0.0)   # OH inconsistent
    rc = nat.lib().mgvae_conv2d_fwd(ctypes.byref(d), None, None, None, None, None)
    assert rc == -1


def test_module_state_dicts_match_reference_manifest(built, golden_dir):
    import json
    from graph.model_with_gan import Model
    from graph.z_discriminator import BarZDiscriminator, PhraseZDiscriminator
    from graph.bar_discriminator_with_feature import BarFeatureDiscriminator
    man = json.load(open(os.path.join(golden_dir, "manifest.json")))
    m = Model()
    want = [["encoder." + n, s] for n, s in man["encoder"]] + [["decoder." + n, s] for n, s in man["decoder"]] + \
           [["phrase_encoder." + n, s] for n, s in man["phrase_encoder"]]
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == want
    for cls, key in ((BarZDiscriminator, "z_discriminator_bar"), (PhraseZDiscriminator, "z_discriminator_phrase"),
                     (BarFeatureDiscriminator, "discriminator_feature")):
        assert [[k, list(v.shape)] for k, v in cls().state_dict().items()] == man[key]
    # D4 statistics of the build's weights_init
    assert abs(m.encoder.layers[0].conv1.weight.mean().item() + 1) < 0.05
    assert m.decoder.layers[3].deConv1.weight.abs().max().item() < 0.1
    assert torch.equal(m.encoder.layers[0].bn.weight, torch.ones(64))


def test_host_helpers_without_a_gpu(built):
    """pure host logic of the step: stacked-latent detection (one discriminator pass over both bar latents), the
    gradient-bucket splitter, the act-mask struct layout of include/mgvae.h"""
    from hipops.train import _stacked
    from hipops.dist import split_buckets
    from hipops import _native as nat
    zz = torch.randn(6, 5)
    a, b = zz[:3], zz[3:]
    s = _stacked(a, b)
    assert s is not None and s.shape == (6, 5) and s.data_ptr() == zz.data_ptr()
    assert _stacked(b, a) is None and _stacked(torch.randn(3, 5), torch.randn(3, 5)) is None
    assert _stacked(zz[:2], zz[2:4]) is None                     # halves must tile the base exactly
    bk = split_buckets(0, 300, 128)                              # from the END backwards, 64-element granularity
    assert bk == [(172, 300), (44, 172), (0, 44)] and split_buckets(3, 3, 128) == []
    assert sorted(bk)[0][0] == 0 and all(a[0] == b[1] for a, b in zip(bk, bk[1:]))
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mgvae.h")).read()
    body = re.search(r"typedef struct MgvaeActMask \{(.*?)\} MgvaeActMask;", hdr, re.S).group(1)
    names = re.findall(r"(\w+)\s*[;,]", body)
    assert names == [f[0] for f in nat.ActMask._fields_], (names, nat.ActMask._fields_)


def _header_prototypes():
    hdr = open(os.path.join(ROOT, "include", "mgvae.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mgvae_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", hdr):
        protos[m.group(2)] = (" ".join(m.group(1).split()), [a.strip() for a in m.group(3).split(",")])
    return protos


def _klass_of_c(arg):
    """argument class of a C parameter declaration: how it travels in a register / an 8-byte word"""
    arg = " ".join(arg.split())
    if arg in ("void", ""):
        return None
    if "*" in arg:
        return "ptr"
    base = arg.rsplit(" ", 1)[0] if " " in arg else arg
    base = base.replace("const ", "").strip()
    return {"int": "i32", "int32_t": "i32", "float": "f32", "double": "f64", "size_t": "u64", "uint64_t": "u64", "long": "i64",
            "unsigned long long": "u64"}[base]


def _klass_of_ctypes(t):
    if t in (ctypes.c_int, ctypes.c_int32):
        return "i32"
    if t is ctypes.c_float:
        return "f32"
    if t is ctypes.c_double:
        return "f64"
    if t in (ctypes.c_size_t, ctypes.c_uint64):
        return "u64"
    if t is ctypes.c_long:
        return "i64"
    return "ptr"


def test_ctypes_table_matches_the_header_argument_by_argument(built):
    """hipops/_native.py::SIGNATURES is what the host layer calls through AND what the chain interpreter's dispatch is
    generated from (tools/gen_chain_dispatch.py): an ``int`` where the header says ``size_t`` would leave the upper half of a
    register undefined.  Every prototype of include/mgvae.h is parsed and compared class by class."""
    from hipops import _native as nat
    protos = _header_prototypes()
    assert set(protos) == set(nat.SIGNATURES), set(protos) ^ set(nat.SIGNATURES)
    for name, (ret, args) in protos.items():
        want = [k for k in (_klass_of_c(a) for a in args) if k is not None]
        res, argtypes = nat.SIGNATURES[name]
        got = [_klass_of_ctypes(t) for t in argtypes]
        assert got == want, (name, got, want)
        want_res = "ptr" if "*" in ret else {"int": "i32", "size_t": "u64"}[ret.replace("const ", "").strip()]
        assert _klass_of_ctypes(res) == want_res, (name, ret)


def test_chain_dispatch_is_current_and_resolves_every_entry_point(built):
    """csrc/chain_dispatch.inc (generated, committed) must be what tools/gen_chain_dispatch.py makes from today's table; the
    built library numbers the entry points the same way; a chain over a cheap host-only entry point runs without a GPU and
    a failing call reports its index."""
    import importlib.util
    import numpy as np
    from hipops import _native as nat
    spec = importlib.util.spec_from_file_location("gen_chain_dispatch", os.path.join(ROOT, "tools", "gen_chain_dispatch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(gen.OUT).read() == gen.render(), "stale csrc/chain_dispatch.inc: run python tools/gen_chain_dispatch.py"
    L = nat.lib()
    names = [n for n, _ in gen.functions()]
    assert L.mgvae_chain_fn_count() == len(names)
    for i, n in enumerate(names):
        assert L.mgvae_chain_fn_id(n.encode()) == i
    assert L.mgvae_chain_fn_id(b"mgvae_no_such_entry_point") == -1
    # two calls of mgvae_set_compute_dtype (host-only state): the second with an invalid code must fail as call 1
    before = L.mgvae_get_compute_dtype()
    fid = L.mgvae_chain_fn_id(b"mgvae_set_compute_dtype")
    calls = (nat.ChainCall * 2)(nat.ChainCall(fid, 1, 0, 0), nat.ChainCall(fid, 1, 1, 0))
    words = np.array([1, 77], dtype=np.uint64)
    failed = ctypes.c_int(-5)
    rc = L.mgvae_chain_run(calls, 2, ctypes.c_void_p(words.ctypes.data), ctypes.byref(failed))
    assert rc == -1 and failed.value == 1 and L.mgvae_get_compute_dtype() == 1
    words[1] = before
    assert L.mgvae_chain_run(calls, 2, ctypes.c_void_p(words.ctypes.data), ctypes.byref(failed)) == 0 and failed.value == -1
    assert L.mgvae_get_compute_dtype() == before
    # wrong argument count for the entry point: rejected, nothing called
    bad = (nat.ChainCall * 1)(nat.ChainCall(fid, 3, 0, 0))
    assert L.mgvae_chain_run(bad, 1, ctypes.c_void_p(words.ctypes.data), ctypes.byref(failed)) == -1 and failed.value == 0
