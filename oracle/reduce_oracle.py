"""CPU oracle for the reduce half -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A plain-Python restatement of what CheckM does between the domtblout text and the QA row, written
from the reference's behaviour (file:line cited per function), quirks included.  PINNED: checked
against goldens produced by importing the reference's own classes in this container
(tools/gen_reduce_golden.py -> tests/golden/reduce_cases.json; tests/test_reduce_oracle.py).
"""
from collections import OrderedDict
import re


def parse_domtblout(text):
    """checkm/hmmer.py:184-200, 255-285: whitespace split, >= 23 tokens, '-' accession -> name.
    A blank line ends the table (IndexError path at hmmer.py:193-200)."""
    rows = []
    for line in text.split('\n'):
        line = line.rstrip()
        if len(line) == 0:
            break
        if line[0] == '#':
            continue
        t = re.split(r'\s+', line)
        if len(t) < 23:
            raise ValueError("Error processing line:\n%s" % line)
        acc = t[4] if t[4] != '-' else t[3]
        rows.append(dict(target_name=t[0], target_accession=t[1], target_length=int(t[2]), query_name=t[3], query_accession=acc,
                         query_length=int(t[5]), full_e_value=float(t[6]), full_score=float(t[7]), full_bias=float(t[8]),
                         dom=int(t[9]), ndom=int(t[10]), c_evalue=float(t[11]), i_evalue=float(t[12]), dom_score=float(t[13]),
                         dom_bias=float(t[14]), hmm_from=int(t[15]), hmm_to=int(t[16]), ali_from=int(t[17]), ali_to=int(t[18]),
                         env_from=int(t[19]), env_to=int(t[20]), acc=float(t[21]), target_description=" ".join(t[22:])))
    return rows


def read_pfam_clans(text):
    """checkm/util/pfam.py:34-56: clan per accession (version stripped), nesting made symmetric."""
    id_to_acc, clan, id_nested = {}, {}, {}
    cur_id = cur_acc = None
    for line in text.split('\n'):
        if '#=GF ID' in line:
            cur_id = line.split()[2].strip()
        elif '#=GF AC' in line:
            cur_acc = line.split()[2].strip()
            cur_acc = cur_acc[0:cur_acc.rfind('.')]
            id_to_acc[cur_id] = cur_acc
        elif '#=GF CL' in line:
            clan[cur_acc] = line.split()[2].strip()
        elif '#=GF NE' in line:
            n = line.split()[2].strip()
            id_nested.setdefault(n, []).append(cur_id)
            id_nested.setdefault(cur_id, []).append(n)
    nested = {}
    for i, ns in id_nested.items():
        nested[id_to_acc[i]] = set(id_to_acc[x] for x in ns)
    return clan, nested


def vet_hit(hit, model, ignore_thresholds, evalue, length, skip_pseudogene):
    """checkm/resultsParser.py:340-377."""
    if not skip_pseudogene:
        if float(hit['ali_to'] - hit['ali_from']) / float(hit['query_length']) < 0.3:
            return False
    ga, tc, nc, acc = model.get('ga'), model.get('tc'), model.get('nc'), model['acc']
    thr = None
    if nc is not None and not ignore_thresholds and 'TIGR' in acc:
        thr = nc
    elif ga is not None and not ignore_thresholds:
        thr = ga
    elif tc is not None and not ignore_thresholds:
        thr = tc
    elif nc is not None and not ignore_thresholds:
        thr = nc
    if thr is not None:
        return thr[0] <= hit['full_score'] and thr[1] <= hit['dom_score']
    if hit['full_e_value'] > evalue:
        return False
    return float(hit['ali_to'] - hit['ali_from']) / float(hit['query_length']) >= length


def add_hits(rows, models, ignore_thresholds, evalue, length, skip_pseudogene):
    """checkm/resultsParser.py:379-399: best domain per (marker, ORF), strict >, survivor moves to the tail."""
    mh = OrderedDict()
    for hit in rows:
        if not vet_hit(hit, models[hit['query_accession']], ignore_thresholds, evalue, length, skip_pseudogene):
            continue
        key = hit['query_accession']
        if key in mh:
            prev = None
            for h in mh[key]:
                if h['target_name'] == hit['target_name']:
                    prev = h
                    break
            if prev is None:
                mh[key].append(hit)
            elif prev['dom_score'] < hit['dom_score']:
                mh[key].append(hit)
                mh[key].remove(prev)
        else:
            mh[key] = [hit]
    return mh


def clan_filter(mh, clan, nested):
    """checkm/util/pfam.py:86-147."""
    out = OrderedDict()
    by_orf = OrderedDict()
    for key, hits in mh.items():
        if key.startswith('PF'):
            for h in hits:
                by_orf.setdefault(h['target_name'], []).append(h)
        else:
            out[key] = hits
    for hits in by_orf.values():
        hits.sort(key=lambda x: (x['full_e_value'], x['i_evalue']))
        dropped = set()
        for i in range(len(hits)):
            if i in dropped:
                continue
            pi = hits[i]['query_accession']
            pi = pi[0:pi.rfind('.')]
            for j in range(i + 1, len(hits)):
                if j in dropped:
                    continue
                pj = hits[j]['query_accession']
                pj = pj[0:pj.rfind('.')]
                if clan.get(pi) == clan.get(pj):
                    sI, eI, sJ, eJ = hits[i]['ali_from'], hits[i]['ali_to'], hits[j]['ali_from'], hits[j]['ali_to']
                    if (sI <= sJ and eI > sJ) or (sJ <= sI and eJ > sI):
                        if not (pi in nested and pj in nested[pi]):
                            dropped.add(j)
        for i in range(len(hits)):
            if i not in dropped:
                out.setdefault(hits[i]['query_accession'], []).append(hits[i])
    return out


def merge_adjacent(mh):
    """checkm/resultsParser.py:401-479: overlap test is irrelevant, every end coordinate takes min()."""
    for hits in mh.values():
        combined = True
        while combined:
            for i in range(len(hits)):
                oi = hits[i]['target_name']
                si = oi[0:oi.rfind('_')]
                combined = False
                jm = None
                for j in range(i + 1, len(hits)):
                    oj = hits[j]['target_name']
                    if si == oj[0:oj.rfind('_')]:
                        try:
                            ni = int(oi[oi.rfind('_') + 1:])
                            nj = int(oj[oj.rfind('_') + 1:])
                        except ValueError:
                            break
                        if abs(ni - nj) == 1:
                            combined = True
                            jm = j
                            break
                if combined:
                    a, b = hits[i], hits[jm]
                    n = dict(a)
                    n['target_name'] = '&&'.join(sorted([oi, b['target_name']]))
                    n['target_length'] = a['target_length'] + b['target_length']
                    for f in ('hmm_from', 'hmm_to', 'ali_from', 'ali_to', 'env_from', 'env_to'):
                        n[f] = min(a[f], b[f])
                    del hits[jm]
                    del hits[i]
                    hits.append(n)
                    break
    return mh


def genome_check(marker_sets, mh, individual):
    """checkm/markerSets.py:206-238."""
    genes = set(m for s in marker_sets for m in s)
    if individual:
        present = multi = 0
        for m in genes:
            if m in mh:
                present += 1
                multi += len(mh[m]) - 1
        n = sum(len(s) for s in marker_sets)
        return 100 * float(present) / n, 100 * float(multi) / n
    comp = cont = 0.0
    for s in marker_sets:
        present = multi = 0
        for m in s:
            c = len(mh.get(m, []))
            if c >= 1:
                present += 1
                multi += c - 1
        comp += float(present) / len(s)
        cont += float(multi) / len(s)
    return 100 * comp / len(marker_sets), 100 * cont / len(marker_sets)


def gene_counts(marker_sets, mh, individual):
    """checkm/resultsParser.py:513-537."""
    hist = [0] * 6
    for m in set(x for s in marker_sets for x in s):
        c = len(mh[m]) if m in mh else 0
        hist[5 if c > 5 else c] += 1
    comp, cont = genome_check(marker_sets, mh, individual)
    return hist + [comp, cont]


def reduce_bin(domtblout_text, models, pfam_text, marker_sets, ignore_thresholds=False, evalue=1e-10, length=0.7,
               skip_pseudogene=False, skip_adj=False, individual=False):
    rows = parse_domtblout(domtblout_text)
    clan, nested = read_pfam_clans(pfam_text)
    mh = add_hits(rows, models, ignore_thresholds, evalue, length, skip_pseudogene)
    mh = clan_filter(mh, clan, nested)
    if not skip_adj:
        mh = merge_adjacent(mh)
    return mh, gene_counts(marker_sets, mh, individual)


def marker_hits_view(mh):
    return [[k, [[h['target_name'], h['target_length'], h['hmm_from'], h['hmm_to'], h['ali_from'], h['ali_to'], h['env_from'], h['env_to'],
                  h['dom_score'], h['full_e_value']] for h in v]] for k, v in mh.items()]
