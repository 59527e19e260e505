"""Small filesystem helpers with the reference's semantics (checkm/common.py:33-44,136-151)."""
import errno
import os
import sys
import logging


def binIdFromFilename(filename):
    """Basename minus one compression suffix minus one extension."""
    binId = os.path.basename(filename)
    if binId.endswith('.gz'):
        binId = binId[0:-3]
    root, _ext = os.path.splitext(binId)
    return root


def makeSurePathExists(path):
    if not path:
        return
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            logging.getLogger('timestamp').error('Specified path could not be created: ' + path)
            sys.exit(1)


def getBinIdsFromOutDir(outDir):
    binIds = []
    binDir = os.path.join(outDir, 'bins')
    for f in os.listdir(binDir):
        if os.path.isdir(os.path.join(binDir, f)):
            binIds.append(f)
    return binIds


def checkFileExists(inputFile):
    if not os.path.exists(inputFile):
        logging.getLogger('timestamp').error('Input file does not exists: ' + inputFile)
        sys.exit(1)


def read_fasta(path):
    """[(name, description, residues)] from a (optionally gzipped) FASTA file."""
    import gzip
    op = gzip.open if path.endswith('.gz') else open
    recs, name, desc, parts = [], None, '', []
    with op(path, 'rt') as f:
        for line in f:
            if not line:
                continue
            if line[0] == '>':
                if name is not None:
                    recs.append((name, desc, ''.join(parts)))
                hdr = line[1:].rstrip('\r\n')
                sp = hdr.split(None, 1)
                name = sp[0] if sp else ''
                desc = sp[1] if len(sp) > 1 else ''
                parts = []
            else:
                parts.append(line.strip())
    if name is not None:
        recs.append((name, desc, ''.join(parts)))
    return recs
