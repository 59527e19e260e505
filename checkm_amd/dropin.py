"""Rebinds CheckM's stage classes to the MI355X implementations (option A)."""


def install():
    import checkm.hmmerModelParser
    import checkm.markerGeneFinder
    import checkm.markerSets
    import checkm.resultsParser
    from checkm_amd import hmmerModelParser as p
    from checkm_amd import markerGeneFinder as g
    from checkm_amd import markerSets as m
    from checkm_amd import resultsParser as r
    checkm.markerGeneFinder.MarkerGeneFinder = g.MarkerGeneFinder
    checkm.resultsParser.ResultsParser = r.ResultsParser
    checkm.resultsParser.ResultsManager = r.ResultsManager
    checkm.markerSets.MarkerSet = m.MarkerSet
    checkm.markerSets.BinMarkerSets = m.BinMarkerSets
    checkm.markerSets.MarkerSetParser = m.MarkerSetParser
    p.HmmModel.__module__ = 'checkm.hmmerModelParser'
    checkm.hmmerModelParser.HmmModel = p.HmmModel
    # every hmmalign call of CheckM (qa --aai_strain, qa -o 9, tree): the masked alignments come from ckm_align
    import checkm.aminoAcidIdentity
    import checkm.hmmerAligner
    from checkm_amd import aminoAcidIdentity as a
    from checkm_amd import hmmerAligner as h
    checkm.hmmerAligner.HmmerAligner = h.HmmerAligner
    checkm.aminoAcidIdentity.AminoAcidIdentity = a.AminoAcidIdentity
    # gene calling in front of the scan: the two translation tables of a bin side by side (same files, same table choice)
    import checkm.prodigal
    from checkm_amd import prodigal as pr
    checkm.prodigal.ProdigalRunner = pr.ProdigalRunner
    checkm.prodigal.ProdigalGeneFeatureParser = pr.ProdigalGeneFeatureParser
