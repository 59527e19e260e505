"""Synthetic inputs of the tests and of bench.py -- profile HMMs with a Pfam-like length distribution and fitted calibration, protein
bins with planted marker genes, the lineage world of configs[2..4], nucleotide genomes for the gene finder.  Generators only: nothing
here is the product's logic, and checkm_amd does not import this package (tests/test_boundary.py)."""
