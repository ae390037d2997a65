"""Allele occurrence counts on MI355X: mirror of the reference's
allele_identification.count_allele_occurence (:129-157); see core_genome.py."""
from __future__ import print_function

from .core_genome import _row_occurrence


def count_allele_occurence(allele_npz_file, ctx=None):
    """Occurrence of each allele over all genomes (reference allele_identification.py:129-157)."""
    df = _row_occurrence(allele_npz_file, 'allele_index', ctx)
    print("\nCounted allele occurence")
    return df
