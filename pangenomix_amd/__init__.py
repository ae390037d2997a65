"""pangenomix_amd: the MI355X (gfx950) hot path of AnnaLew/pangenomix.

Pan-genome construction (greedy sequence clustering with cd-hit's rules, gene x genome
presence/absence tables) and pan/core rarefaction curves, behind the reference's own
Python entry points. Host code is Python; the arithmetic lives in hand-written HIP
kernels behind the C ABI of include/pgx.h (pangenomix_amd/libpgx.so).

    from pangenomix_amd import pangenome, pangenome_analysis, sparse_utils
"""
from . import sparse_utils, pangenome, pangenome_analysis  # noqa: F401

__version__ = '0.1.0'
