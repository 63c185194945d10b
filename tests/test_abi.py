"""The C-ABI library: it loads, exports exactly what include/plonk_hip.h declares, its host-only
encoders match the oracle, and it fails loudly (no fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from oracle import curve, protocol as P
from oracle.field import R_MOD


def _declared():
    text = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plonk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from plonk_amd import _lib
    names = _declared()
    assert len(names) >= 30
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "missing export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_struct_layouts():
    from plonk_amd import _lib
    assert C.sizeof(_lib.Fr) == 32 and C.sizeof(_lib.G1) == 104
    assert C.sizeof(_lib.Proof) == 9 * 104 + 6 * 32


def test_host_encoders_match_oracle():
    from plonk_amd import _lib, api
    for v in (0, 1, 5, R_MOD - 1, 12345678901234567890):
        out = (C.c_uint8 * 32)()
        s = api._fr_struct(v)
        assert _lib.lib.plonk_fr_to_bytes(C.byref(s), out) == 0
        assert bytes(out) == v.to_bytes(32, "little")
        back = _lib.Fr()
        assert _lib.lib.plonk_fr_from_bytes(out, C.byref(back)) == 0
        assert api._fr_value(back) == v
    bad = (C.c_uint8 * 32)(*R_MOD.to_bytes(32, "little"))
    assert _lib.lib.plonk_fr_from_bytes(bad, C.byref(_lib.Fr())) == _lib.PLONK_E_INVALID
    G = curve.G1_GEN
    for pt in (G, curve.affine_neg(G), curve.to_affine(curve.scalar_mul(G, 7)), None):
        out = (C.c_uint8 * 48)()
        g = api._g1_struct(pt)
        assert _lib.lib.plonk_g1_compress(C.byref(g), out) == 0
        assert bytes(out) == curve.compress(pt)
        assert api._g1_value(g) == pt


def test_proof_serialize_and_challenges_match_oracle():
    from plonk_amd import _lib, api
    from conftest import golden
    e = [x for x in golden("proofs.json") if x["name"] == "c2_range_proof_255"][0]
    wire = bytes.fromhex(e["proof"])
    raw = _lib.Proof()
    off = 0
    for k in P.PROOF_G1[:7]:
        setattr(raw, k, api._g1_struct(curve.decompress(wire[off:off + 48]))); off += 48
    for k in P.PROOF_FR:
        setattr(raw, k, api._fr_struct(int.from_bytes(wire[off:off + 32], "little"))); off += 32
    for k in P.PROOF_G1[7:]:
        setattr(raw, k, api._g1_struct(curve.decompress(wire[off:off + 48]))); off += 48
    pr = api.Proof(raw)
    assert pr.to_bytes() == wire
    assert pr.challenges([]) == [int(c, 16) for c in e["challenges"]]


def test_no_gpu_fails_loudly():
    """Without a HIP device the product must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import plonk_amd
    with pytest.raises(plonk_amd.PlonkError):
        plonk_amd.Context(0)
    with pytest.raises(plonk_amd.PlonkError):
        plonk_amd.fft([1, 2, 3, 4], plonk_amd.api.root_of_unity(4), 4)


def test_product_does_not_import_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle|oracle/|libplonk_ref", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plonk_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cuh", "Makefile")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
