#!/usr/bin/env python3
"""Extract the low-pass tap literals of the reference's discrete wavelets.

The reference (Prophetizo/VectorWave) keeps its filter banks as Java `double[]`
literals.  They are data, not code: this script reads the literal numbers as
text, evaluates them as IEEE-754 doubles (Python's float() and Java's decimal
literal conversion are both correctly rounded, so the bits agree) and writes
them, as C99 hex floats, to vectorwave_amd/wavelet_tables.json.  The derived
filters (high-pass QMF, reconstruction pairs) are NOT stored; they are rebuilt
from these low-pass tables by vectorwave_amd/wavelets.py and by the C++ host
mirror with the same arithmetic as the reference.

Sources (reference file:line):
  api/Haar.java:40-44                    Haar  (1.0 / Math.sqrt(2))
  api/Daubechies.java:58-160             DB2..DB10 literals
  util/DaubechiesGenerator.java:34-324   DB12..DB38 literals + normalisation
  api/Symlet.java:53-434                 SYM2..SYM20
  api/Coiflet.java:38-568                COIF1..COIF17
  api/DiscreteMeyer.java:51-68           DMEY
  api/BiorthogonalSpline.java:21-255     BIORx.y (decomposition + reconstruction low-pass)

Run from the repo root, in the build container only (needs /root/reference):
    python tools/extract_wavelet_tables.py
"""
import json
import math
import os
import re
import sys

REF = "/root/reference/src/main/java/ai/prophetizo/wavelet"
OUT = os.path.join(os.path.dirname(__file__), "..", "vectorwave_amd", "wavelet_tables.json")

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return src


def parse_array(body, consts=None):
    """Evaluate the comma-separated items of a Java double[] initialiser."""
    out = []
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if re.fullmatch(NUM, item):
            out.append(float(item))
            continue
        neg = item.startswith("-")
        name = item[1:].strip() if neg else item
        if consts is None or name not in consts:
            raise ValueError("unparsed array item %r" % item)
        out.append(-consts[name] if neg else consts[name])
    return out


def daubechies():
    res = {}
    src = strip_comments(open(f"{REF}/api/Daubechies.java").read())
    for m in re.finditer(r'new Daubechies\(\s*"(db\d+)",\s*new double\[\]\s*\{(.*?)\}', src, flags=re.S):
        res[m.group(1)] = parse_array(m.group(2))
    gen = strip_comments(open(f"{REF}/util/DaubechiesGenerator.java").read())
    for m in re.finditer(r"case (\d+) -> \{\s*yield new double\[\]\s*\{(.*?)\};", gen, flags=re.S):
        res["db" + m.group(1)] = normalize_daubechies(parse_array(m.group(2)))
    return res


def normalize_daubechies(raw):
    """util/DaubechiesGenerator.java:288-324, same operation order."""
    tolerance = 1e-10
    energy = 0.0
    total = 0.0
    for c in raw:
        energy += c * c
        total += c
    if abs(energy - 1.0) < tolerance and abs(total - math.sqrt(2)) < tolerance:
        return list(raw)
    energy_scale = 1.0 / math.sqrt(energy)
    normalized = [c * energy_scale for c in raw]
    total = 0.0
    for c in normalized:
        total += c
    sum_scale = math.sqrt(2) / total
    return [c * sum_scale for c in normalized]


def simple_family(path, ctor, prefix):
    res = {}
    src = strip_comments(open(path).read())
    for m in re.finditer(r"new %s\((\d+),\s*new double\[\]\s*\{(.*?)\}\)" % ctor, src, flags=re.S):
        res[prefix + m.group(1)] = parse_array(m.group(2))
    return res


def dmey():
    src = strip_comments(open(f"{REF}/api/DiscreteMeyer.java").read())
    m = re.search(r"DMEY_62_COEFFICIENTS = new double\[\]\s*\{(.*?)\};", src, flags=re.S)
    return {"dmey": parse_array(m.group(1))}


def biorthogonal():
    """Returns name -> (lowPassDecomp, lowPassRecon)."""
    s2 = math.sqrt(2.0)
    consts = {  # api/BiorthogonalSpline.java:21-29
        "ONE_EIGHTH": 1.0 / 8.0,
        "MINUS_ONE_EIGHTH": -1.0 / 8.0,
        "ONE": 1.0,
        "SQRT_2_INVERSE": 1.0 / s2,
        "SQRT_2": s2,
        "HALF_SQRT_2_INVERSE": 0.5 / s2,
        "QUARTER_SQRT_2_INVERSE": 0.25 / s2,
        "THREE_QUARTER_SQRT_2_INVERSE": 0.75 / s2,
        "THREE_HALF_SQRT_2_INVERSE": 1.5 / s2,
    }
    res = {}
    src = strip_comments(open(f"{REF}/api/BiorthogonalSpline.java").read())
    pat = (r'new BiorthogonalSpline\(\s*"(bior\d\.\d)",\s*\d+,\s*\d+,\s*'
           r"new double\[\]\s*\{(.*?)\},\s*new double\[\]\s*\{(.*?)\}")
    for m in re.finditer(pat, src, flags=re.S):
        res[m.group(1)] = (parse_array(m.group(2), consts), parse_array(m.group(3), consts))
    return res


def build_tables():
    """The JSON document, extracted from the reference tree in memory."""
    ortho = {"haar": [1.0 / math.sqrt(2), 1.0 / math.sqrt(2)]}
    ortho.update(daubechies())
    ortho.update(simple_family(f"{REF}/api/Symlet.java", "Symlet", "sym"))
    ortho.update(simple_family(f"{REF}/api/Coiflet.java", "Coiflet", "coif"))
    ortho.update(dmey())
    bior = biorthogonal()
    return {
        "_comment": "Low-pass tap tables as C99 hex floats; generated by tools/extract_wavelet_tables.py",
        "orthogonal": {k: [x.hex() for x in v] for k, v in ortho.items()},
        "biorthogonal": {k: {"dec_lo": [x.hex() for x in v[0]], "rec_lo": [x.hex() for x in v[1]]}
                         for k, v in bior.items()},
    }


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; the committed JSON is authoritative")
    doc = build_tables()
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
    print("orthogonal:", len(doc["orthogonal"]), "biorthogonal:", len(doc["biorthogonal"]))
    for k, v in doc["orthogonal"].items():
        print("  %-7s L=%d" % (k, len(v)))
    for k, v in doc["biorthogonal"].items():
        print("  %-8s Ldec=%d Lrec=%d" % (k, len(v["dec_lo"]), len(v["rec_lo"])))


if __name__ == "__main__":
    main()
