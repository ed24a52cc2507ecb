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
    dm = re.search(r"dimensions\s+[^;]*?nchar\s*=\s*(\d+)", text_nc, flags=re.I)
    nchar = int(dm.group(1)) if dm else None
    fm = re.search(r"\bformat\b([^;]*);", text_nc, flags=re.I | re.S)
    fmt = fm.group(1) if fm else ""
    interleaved = re.search(r"\binterleave(\s*=\s*yes)?\b", fmt, flags=re.I) is not None and \
        re.search(r"\binterleave\s*=\s*no\b", fmt, flags=re.I) is None
    names, rows = _matrix_by_lines(m.group(1))
    if not rows:
        raise NexusError("%s has an empty MATRIX block" % path)
    if nchar is not None and not interleaved and any(len(r) != nchar for r in rows):
        # a sequential (non-interleaved) matrix may wrap a sequence over several lines: read it as a token stream
        names, rows = _matrix_by_tokens(m.group(1), nchar)
    n = len(rows[0])
    for name, r in zip(names, rows):
        if len(r) != n:  # tapir/compute.py:103 asserts equal lengths
            raise NexusError("sequence %s has %d characters, expected %d" % (name, len(r), n))
    if nchar is not None and nchar != n:
        raise NexusError("%s: nchar=%s but sequences have %d characters" % (path, nchar, n))
    mc = re.search(r"matchchar\s*=\s*(\S)", fmt, flags=re.I)
    if mc:  # MATCHCHAR: "same state as the first sequence at this position"
        ch = mc.group(1)
        first = rows[0]
        rows = [first] + ["".join(f if c == ch else c for c, f in zip(r, first)) for r in rows[1:]]
    return names, rows


def _split_label(line):
    if line[0] in "'\"":
        q = line[0]
        end = line.index(q, 1)
        return line[1:end], line[end + 1:]
    parts = line.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def _matrix_by_lines(block):
    """One `label sequence-chunk` per line; a label that comes back continues its sequence (interleaved blocks)."""
    names, seqs = [], {}
    for line in block.splitlines():
        line = line.strip()
        if not line:
            continue
        name, rest = _split_label(line)
        if name not in seqs:
            names.append(name)
            seqs[name] = []
        seqs[name].append(re.sub(r"\s+", "", rest))
    return names, ["".join(seqs[n]) for n in names]


def _matrix_by_tokens(block, nchar):
    """Sequential format: a label, then characters (over as many lines and blanks as it takes) until nchar are read."""
    names, rows = [], []
    text = block.strip()
    pos = 0
    while pos < len(text):
        while pos < len(text) and text[pos].isspace():
            pos += 1
        if pos >= len(text):
            break
        name, rest = _split_label(text[pos:])
        pos = len(text) - len(rest)
        chars = []
        while len(chars) < nchar and pos < len(text):
            c = text[pos]
            pos += 1
            if not c.isspace():
                chars.append(c)
        names.append(name)
        rows.append("".join(chars))
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
