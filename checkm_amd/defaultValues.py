"""Constants and file names of the marker-gene path (mirror of checkm/defaultValues.py:34-104,
restricted to what the scan/reduce path uses).  The data root comes from CHECKM_DATA_PATH, as in
checkm/checkmData.py:115-121."""
import os


class DefaultValues(object):
    MARKERS_TO_EXCLUDE = {'TIGR00398', 'TIGR00399'}

    E_VAL = 1e-10
    LENGTH = 0.7
    PSEUDOGENE_LENGTH = 0.3

    TAXON_MARKER_FILE_HEADER = '# [Taxon Marker File]'
    LINEAGE_MARKER_FILE_HEADER = '# [Lineage Marker File]'
    SEQ_CONCAT_CHAR = '&&'

    CHECKM_DATA_DIR = os.environ.get('CHECKM_DATA_PATH', '')
    PHYLO_HMM_MODELS = os.path.join(CHECKM_DATA_DIR, 'hmms', 'phylo.hmm')
    HMM_MODELS = os.path.join(CHECKM_DATA_DIR, 'hmms', 'checkm.hmm')
    PFAM_CLAN_FILE = os.path.join(CHECKM_DATA_DIR, 'pfam', 'Pfam-A.hmm.dat')
    SELECTED_MARKER_SETS = os.path.join(CHECKM_DATA_DIR, 'selected_marker_sets.tsv')
    TAXON_MARKER_SETS = os.path.join(CHECKM_DATA_DIR, 'taxon_marker_sets.tsv')

    PHYLO_HMM_MODEL_INFO = 'phylo_hmm_info.pkl.gz'
    CHECKM_HMM_MODEL_INFO = 'checkm_hmm_info.pkl.gz'
    HMMER_TABLE_PHYLO_OUT = 'hmmer.tree.txt'
    HMMER_PHYLO_OUT = 'hmmer.tree.ali.txt'
    HMMER_TABLE_OUT = 'hmmer.analyze.txt'
    HMMER_OUT = 'hmmer.analyze.ali.txt'
    PRODIGAL_AA = 'genes.faa'
    PRODIGAL_NT = 'genes.fna'
    PRODIGAL_GFF = 'genes.gff'
    BIN_STATS_PHYLO_OUT = 'bin_stats.tree.tsv'
    BIN_STATS_OUT = 'bin_stats.analyze.tsv'
    BIN_STATS_EXT_OUT = 'bin_stats_ext.tsv'
    MARKER_GENE_STATS = 'marker_gene_stats.tsv'

    @classmethod
    def set_data_root(cls, root):
        """Re-point every data path (tests; the reference fixes them at import time)."""
        cls.CHECKM_DATA_DIR = root
        cls.PHYLO_HMM_MODELS = os.path.join(root, 'hmms', 'phylo.hmm')
        cls.HMM_MODELS = os.path.join(root, 'hmms', 'checkm.hmm')
        cls.PFAM_CLAN_FILE = os.path.join(root, 'pfam', 'Pfam-A.hmm.dat')
        cls.SELECTED_MARKER_SETS = os.path.join(root, 'selected_marker_sets.tsv')
        cls.TAXON_MARKER_SETS = os.path.join(root, 'taxon_marker_sets.tsv')
