"""NEXUS DNA matrix reader -> state masks (replaces DendroPy's DnaCharacterMatrix and HyPhy's ReadDataFile).

Reference call sites: tapir/compute.py:100 (`dendropy.DnaCharacterMatrix.get_from_path(alignment,'nexus')`)
and models_and_rates.bf:908-910 (`ReadDataFile(FILE_NUCLEOTIDES)`).

State masks (one byte per cell): A=1, C=2, G=4, T/U=8, IUPAC ambiguity codes are unions, gap / missing /
N / X = 15.  The `format` line's `missing=` and `gap=` symbols are honoured.
"""
import re

import numpy as np

IUPAC = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
         "B": 14, "D": 13, "H": 11, "V": 7, "N": 15, "X": 15, "?": 15, "-": 15, ".": 15}

_LUT = np.zeros(256, dtype=np.uint8)
for _k, _v in IUPAC.items():
    _LUT[ord(_k)] = _v
    _LUT[ord(_k.lower())] = _v


class NexusError(ValueError):
    pass


def read_matrix(path):
    """Returns (names, seqs): taxon labels and their sequences (strings), interleaved blocks joined."""
    with open(path) as fh:
        text = fh.read()
    text_nc = re.sub(r"\[[^\]]*\]", "", text)
    if not text_nc.strip():
        raise NexusError("%s is empty" % path)
    m = re.search(r"\bmatrix\b(.*?);", text_nc, flags=re.I | re.S)
    if not m:
        raise NexusError("%s has no MATRIX block" % path)
    names, seqs = [], {}
    for line in m.group(1).splitlines():
        line = line.strip()
        if not line:
            continue
        if line[0] in "'\"":
            q = line[0]
            end = line.index(q, 1)
            name, rest = line[1:end], line[end + 1:]
        else:
            parts = line.split(None, 1)
            name, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        if name not in seqs:
            names.append(name)
            seqs[name] = []
        seqs[name].append(re.sub(r"\s+", "", rest))
    rows = ["".join(seqs[n]) for n in names]
    if not rows:
        raise NexusError("%s has an empty MATRIX block" % path)
    n = len(rows[0])
    for name, r in zip(names, rows):
        if len(r) != n:  # tapir/compute.py:103 asserts equal lengths
            raise NexusError("sequence %s has %d characters, expected %d" % (name, len(r), n))
    dm = re.search(r"dimensions\s+[^;]*?nchar\s*=\s*(\d+)", text_nc, flags=re.I)
    if dm and int(dm.group(1)) != n:
        raise NexusError("%s: nchar=%s but sequences have %d characters" % (path, dm.group(1), n))
    return names, rows


def format_symbols(path):
    """(missing, gap) symbols of the FORMAT line, defaults '?' and '-'."""
    with open(path) as fh:
        text = fh.read()
    miss = re.search(r"missing\s*=\s*(\S)", text, flags=re.I)
    gap = re.search(r"gap\s*=\s*(\S)", text, flags=re.I)
    return (miss.group(1) if miss else "?"), (gap.group(1) if gap else "-")


def encode(rows, missing="?", gap="-"):
    """Sequences -> uint8 [ntaxa, ncols] state masks."""
    lut = _LUT.copy()
    lut[ord(missing)] = 15
    lut[ord(gap)] = 15
    out = np.empty((len(rows), len(rows[0])), dtype=np.uint8)
    for i, r in enumerate(rows):
        codes = lut[np.frombuffer(r.encode("ascii"), dtype=np.uint8)]
        if (codes == 0).any():
            bad = r[int(np.argmax(codes == 0))]
            raise NexusError("unknown DNA character %r" % bad)
        out[i] = codes
    return out


def read_states(path):
    """(names, states uint8 [ntaxa, ncols]) for one alignment file."""
    names, rows = read_matrix(path)
    miss, gap = format_symbols(path)
    return names, encode(rows, miss, gap)


def base_frequencies_from_histogram(hist):
    """HarvestFrequencies(Freqs, filter, 1, 1, 1) (models_and_rates.bf:968) from a [.., 16] histogram of
    state masks: every cell adds 1/popcount(mask) to each base it may be, so a gap counts 1/4 to each
    base (reproduces the fixture's 0.14/0.19/0.33/0.34)."""
    hist = np.asarray(hist, dtype=np.float64)
    w = np.zeros((16, 4))
    for mask in range(1, 16):
        bits = [(mask >> k) & 1 for k in range(4)]
        w[mask] = np.array(bits) / sum(bits)
    w[0] = 0.25  # an all-zero code is treated as missing
    counts = hist @ w
    return counts / counts.sum(axis=-1, keepdims=True)
