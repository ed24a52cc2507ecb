"""tapir_amd -- MI355X-native site-rate + phylogenetic-informativeness engine behind tapir's own surface.

Scope: the one data-parallel hot path of faircloth-lab/tapir (bin/tapir_compute.py:84-123 `worker` and
what it calls), see DESIGN.md.  Submodules:

  engine    ctypes binding of libtphip.so (include/tphip.h); no CPU fallback
  newick    tree reader/writer (replaces DendroPy for this path)
  nexus     NEXUS DNA matrix reader -> packed state masks
  compute   mirror of tapir/compute.py's function names on top of the engine
  base      mirror of tapir/base.py's argparse helpers
  db        mirror of tapir/db.py (byte-identical DDL)
  pipeline  all loci at once: the batch replacement of Pool.map(worker, params)
  dist      loci sharded round-robin over ranks + one all-gather of PI tables
  synth     seeded synthetic alignments of the BASELINE.json shapes
"""
__version__ = "0.1.0"
