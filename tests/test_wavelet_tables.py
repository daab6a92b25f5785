"""Guards on the filter-tap tables (vectorwave_amd/wavelet_tables.json).

Oracle and GPU library read the same table file, so no parity test can notice an edit to it.  Three guards:
  * when the reference tree is present, the extraction (tools/extract_wavelet_tables.py: Java double[] literals read
    as text) is re-run in memory and must equal the committed file bit for bit;
  * always, the file's canonical content must have the digest recorded here (an edit has to be deliberate);
  * always, the reference's own coefficient checks are replayed on every orthogonal table: sum = sqrt(2), sum of
    squares = 1 and zero correlation at even shifts, at the reference's tolerances (api/Daubechies.java:348-379:
    1e-10; api/Symlet.java:482-492: sym8 1e-6, sym10 2e-4; api/Coiflet.java:668-670: coif2 1e-4;
    src/test/.../verification/WaveletCoefficientVerificationTest.java:37-42: dmey 3e-3), the QMF rule of the
    high-pass (api/Daubechies.java:320-327) and the wavelet's vanishing moments for the Daubechies family
    (:381-394)."""
import hashlib
import importlib.util
import json
import math
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = os.path.join(ROOT, "vectorwave_amd", "wavelet_tables.json")
DIGEST = "9f8e32ab011a2e5656894ec5431b75d21b9a0a3ab61fe01fcf80faa010c6175d"
KNOWN_TOLERANCES = {"sym8": 1e-6, "sym10": 2e-4, "coif2": 1e-4, "dmey": 3e-3}


def _doc():
    return json.load(open(TABLES))


def test_committed_tables_have_the_recorded_digest():
    d = _doc()
    canon = json.dumps({k: d[k] for k in ("orthogonal", "biorthogonal")}, sort_keys=True, separators=(",", ":"))
    assert hashlib.sha256(canon.encode()).hexdigest() == DIGEST, \
        "wavelet_tables.json changed: re-run tools/extract_wavelet_tables.py against the reference and update DIGEST"
    assert len(d["orthogonal"]) == 57 and len(d["biorthogonal"]) == 15


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/main/java"), reason="reference tree not present")
def test_tables_equal_a_fresh_extraction_from_the_reference():
    spec = importlib.util.spec_from_file_location("extract_wavelet_tables", os.path.join(ROOT, "tools", "extract_wavelet_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.build_tables()
    have = _doc()
    assert set(fresh["orthogonal"]) == set(have["orthogonal"])
    assert set(fresh["biorthogonal"]) == set(have["biorthogonal"])
    for name, taps in fresh["orthogonal"].items():
        assert taps == have["orthogonal"][name], name  # hex floats: equal strings are equal bits
    for name, pair in fresh["biorthogonal"].items():
        assert pair == have["biorthogonal"][name], name


@pytest.mark.parametrize("name", sorted(_doc()["orthogonal"]))
def test_orthogonal_table_passes_the_reference_checks(name):
    h = [float.fromhex(x) for x in _doc()["orthogonal"][name]]
    tol = KNOWN_TOLERANCES.get(name, 1e-10)
    assert len(h) % 2 == 0
    total = 0.0
    for c in h:
        total += c
    assert abs(total - math.sqrt(2)) <= tol, "sum of taps"
    squares = 0.0
    for c in h:
        squares += c * c
    assert abs(squares - 1.0) <= tol, "sum of squared taps"
    for k in range(2, len(h), 2):
        dot = 0.0
        for n in range(len(h) - k):
            dot += h[n] * h[n + k]
        assert abs(dot) <= tol, "correlation at shift %d" % k


def test_high_pass_is_the_alternating_flip_and_kills_polynomials():
    """The library's derived filters: g[i] = (-1)^i h[L-1-i]; for dbN the wavelet filter has N vanishing moments at
    the reference's growing tolerance 1e-10 * 10^p."""
    from vectorwave_amd.wavelets import get_wavelet
    for name in sorted(_doc()["orthogonal"]):
        w = get_wavelet(name)
        h, g = list(w.dec_lo), list(w.dec_hi)
        L = len(h)
        for i in range(L):
            want = h[L - 1 - i] if i % 2 == 0 else -h[L - 1 - i]
            assert g[i] == want, "%s g[%d]" % (name, i)
    orders = [int(n[2:]) for n in _doc()["orthogonal"] if n.startswith("db") and int(n[2:]) <= 10]
    assert 4 in orders and 10 in orders
    for order in orders:
        g = list(get_wavelet("db%d" % order).dec_hi)
        for p in range(order):
            moment = 0.0
            for n in range(len(g)):
                moment += math.pow(n, p) * g[n]
            assert abs(moment) <= 1e-10 * math.pow(10, p), "db%d moment %d = %g" % (order, p, moment)


def test_db4_literals():
    """Spot values any reader can check against the literature (Daubechies 1992, table 6.1, D8 scaled to sum sqrt 2)."""
    h = [float.fromhex(x) for x in _doc()["orthogonal"]["db4"]]
    assert len(h) == 8
    assert abs(h[0] - 0.2303778133088964) < 1e-12 and abs(h[1] - 0.7148465705529154) < 1e-12
    assert abs(h[7] + 0.0105974017850690) < 1e-12
