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


_RE_COMMENT = re.compile(r"\[[^\]]*\]")
_RE_MATRIX = re.compile(r"\bmatrix\b", flags=re.I)
_RE_NCHAR = re.compile(r"dimensions\s+[^;]*?nchar\s*=\s*(\d+)", flags=re.I)
_RE_FORMAT = re.compile(r"\bformat\b([^;]*);", flags=re.I | re.S)
_RE_INTERLEAVE = re.compile(r"\binterleave(\s*=\s*yes)?\b", flags=re.I)
_RE_INTERLEAVE_NO = re.compile(r"\binterleave\s*=\s*no\b", flags=re.I)
_RE_MATCHCHAR = re.compile(r"matchchar\s*=\s*(\S)", flags=re.I)
_RE_MISSING = re.compile(r"missing\s*=\s*(\S)", flags=re.I)
_RE_GAP = re.compile(r"gap\s*=\s*(\S)", flags=re.I)


def _read_text(path):
    with open(path) as fh:
        return fh.read()


def read_matrix(path, text=None):
    """Returns (names, seqs): taxon labels and their sequences (strings), interleaved blocks joined."""
    if text is None:
        text = _read_text(path)
    text_nc = _RE_COMMENT.sub("", text) if "[" in text else text
    if not text_nc.strip():
        raise NexusError("%s is empty" % path)
    # the MATRIX block: from the keyword to the next semicolon (found with str.find: a lazy regex over a 64 KB block
    # was a third of the parse time)
    m = _RE_MATRIX.search(text_nc)
    end = text_nc.find(";", m.end()) if m else -1
    if not m or end < 0:
        raise NexusError("%s has no MATRIX block" % path)
    block = text_nc[m.end():end]
    header = text_nc[:m.start()]
    dm = _RE_NCHAR.search(header) or _RE_NCHAR.search(text_nc, end)
    nchar = int(dm.group(1)) if dm else None
    fm = _RE_FORMAT.search(header) or _RE_FORMAT.search(text_nc, end)
    fmt = fm.group(1) if fm else ""
    interleaved = _RE_INTERLEAVE.search(fmt) is not None and _RE_INTERLEAVE_NO.search(fmt) is None
    names, rows = _matrix_by_lines(block)
    if not rows:
        raise NexusError("%s has an empty MATRIX block" % path)
    if nchar is not None and not interleaved and any(len(r) != nchar for r in rows):
        # a sequential (non-interleaved) matrix may wrap a sequence over several lines: read it as a token stream
        names, rows = _matrix_by_tokens(block, nchar)
    n = len(rows[0])
    for name, r in zip(names, rows):
        if len(r) != n:  # tapir/compute.py:103 asserts equal lengths
            raise NexusError("sequence %s has %d characters, expected %d" % (name, len(r), n))
    if nchar is not None and nchar != n:
        raise NexusError("%s: nchar=%s but sequences have %d characters" % (path, nchar, n))
    mc = _RE_MATCHCHAR.search(fmt)
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
        seqs[name].append("".join(rest.split()))      # = re.sub(r"\s+", "", rest), several times faster
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


def format_symbols(path, text=None):
    """(missing, gap) symbols of the FORMAT line, defaults '?' and '-'."""
    if text is None:
        text = _read_text(path)
    m = _RE_MATRIX.search(text)          # the symbols are declared before the matrix; its 64 KB need not be searched
    miss = _RE_MISSING.search(text, 0, m.start()) if m else None
    gap = _RE_GAP.search(text, 0, m.start()) if m else None
    if m and (miss is None or gap is None):   # ... unless a file declares them after it
        end = text.find(";", m.end())
        if end >= 0:
            miss = miss or _RE_MISSING.search(text, end)
            gap = gap or _RE_GAP.search(text, end)
    if not m:
        miss, gap = _RE_MISSING.search(text), _RE_GAP.search(text)
    return (miss.group(1) if miss else "?"), (gap.group(1) if gap else "-")


def encode(rows, missing="?", gap="-"):
    """Sequences -> uint8 [ntaxa, ncols] state masks."""
    lut = _LUT.copy()
    lut[ord(missing)] = 15
    lut[ord(gap)] = 15
    n = len(rows[0])
    if any(len(r) != n for r in rows):
        raise NexusError("sequences of unequal length")
    try:
        raw = np.frombuffer("".join(rows).encode("ascii"), dtype=np.uint8)
    except UnicodeEncodeError:
        bad = next(c for r in rows for c in r if ord(c) > 127)
        raise NexusError("unknown DNA character %r" % bad)
    out = lut[raw].reshape(len(rows), n)       # one table look-up for the whole alignment
    if n and not out.all():
        i, j = np.argwhere(out == 0)[0]
        raise NexusError("unknown DNA character %r" % rows[int(i)][int(j)])
    return out


def read_states(path):
    """(names, states uint8 [ntaxa, ncols]) for one alignment file."""
    text = _read_text(path)
    names, rows = read_matrix(path, text)
    miss, gap = format_symbols(path, text)
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
