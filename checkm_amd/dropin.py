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
