"""Header view of HMMER3 profiles, API-compatible with checkm/hmmerModelParser.py:27-83.

`HmmModel` objects carry name/acc/leng/ga/tc/nc and pickle like the reference's
(checkm/markerSets.py:524-540).  The reference's header skim never resets its key dict between
records (hmmerModelParser.py:56), so ACC/GA/TC/NC of one record leak into following records that
lack them; `sticky_models` reproduces exactly that from the library's raw per-record headers.
"""


class HmmModelError(Exception):
    pass


class HmmModel(object):
    """Threshold/header view of one profile."""

    def __init__(self, keys):
        self.ga = None
        self.tc = None
        self.nc = None
        if 'acc' not in keys:
            self.acc = keys['name']
        for k, v in keys.items():
            setattr(self, k, v)


def sticky_models(headers):
    """headers: raw per-record dicts (name, acc|None, leng, ga|None, tc|None, nc|None), file order.
    Yields HmmModel with the carry-over of hmmerModelParser.py:56-81."""
    carried = {}
    for hd in headers:
        carried['format'] = 'HMMER3/f'
        carried['name'] = hd['name']
        if hd.get('acc') is not None:
            carried['acc'] = hd['acc']
        carried['leng'] = int(hd['leng'])
        for t in ('ga', 'tc', 'nc'):
            if hd.get(t) is not None:
                carried[t] = (float(hd[t][0]), float(hd[t][1]))
        yield HmmModel(dict(carried))


def models_dict(headers):
    """{acc: HmmModel} keyed the way HmmModelParser.models() keys it (hmmerModelParser.py:46-52)."""
    out = {}
    for m in sticky_models(headers):
        out[m.acc] = m
    return out


def read_headers(hmm_file):
    """Raw header records of a HMMER3 text file (host-side text skim; no scores are read here)."""
    headers = []
    cur = None
    in_body = False
    with open(hmm_file) as f:
        for line in f:
            if in_body:
                if line.startswith('//'):
                    headers.append(cur)
                    cur, in_body = None, False
                continue
            if line.startswith('HMMER'):
                cur = {'name': None, 'acc': None, 'leng': None, 'ga': None, 'tc': None, 'nc': None}
            elif line.startswith('HMM'):
                in_body = True
            else:
                fields = line.rstrip().split(None, 1)
                if len(fields) != 2:
                    raise HmmModelError("malformed header line: %r" % line)
                tag, val = fields
                if cur is None:
                    raise HmmModelError("header line outside a record: %r" % line)
                if tag == 'NAME':
                    cur['name'] = val
                elif tag == 'ACC':
                    cur['acc'] = val
                elif tag == 'LENG':
                    cur['leng'] = int(val)
                elif tag in ('GA', 'TC', 'NC'):
                    p = val.split()
                    if len(p) != 2:
                        raise HmmModelError("malformed %s line" % tag)
                    cur[tag.lower()] = (float(p[0].replace(';', '')), float(p[1].replace(';', '')))
    return headers


class HmmModelParser(object):
    """Drop-in for checkm.hmmerModelParser.HmmModelParser (header view only)."""

    def __init__(self, hmmFile):
        self.hmmFile = hmmFile
        self._headers = read_headers(hmmFile)

    def models(self):
        return models_dict(self._headers)

    def simpleParse(self):
        return sticky_models(self._headers)

    def parse(self):
        return sticky_models(self._headers)
