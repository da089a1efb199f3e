"""Minimal FITS binary-table reader / writer for the reference's on-disk cache formats:
healpy write_alm / read_alm (columns index = l^2 + l + m + 1, real, imag) and write_map / read_map.
No astropy / healpy is available, so the FITS standard (80-char cards, 2880-byte blocks, big-endian data)
is written directly.  Files written here are readable by healpy and vice versa for these two layouts."""
import gzip
import os

import numpy as np

_BLOCK = 2880


def _card(key, value=None, comment=''):
    if value is None:
        s = key
    elif isinstance(value, bool):
        s = '%-8s= %20s' % (key, 'T' if value else 'F')
    elif isinstance(value, (int, np.integer)):
        s = '%-8s= %20d' % (key, value)
    elif isinstance(value, float):
        s = '%-8s= %20.14G' % (key, value)
    else:
        s = "%-8s= '%-8s'" % (key, str(value))
    if comment:
        s += ' / ' + comment
    return s[:80].ljust(80)


def _header(cards):
    txt = ''.join(cards) + 'END'.ljust(80)
    pad = (-len(txt)) % _BLOCK
    return (txt + ' ' * pad).encode('ascii')


def write_bintable(fname, columns, extname='xtension', extra=(), overwrite=True):
    """columns: list of (name, 1-D or 2-D array); 2-D arrays give vector columns (repeat = shape[1])."""
    if os.path.exists(fname) and not overwrite:
        raise OSError('%s exists' % fname)
    nrows = columns[0][1].shape[0]
    fields, tforms = [], []
    for name, arr in columns:
        arr = np.asarray(arr)
        assert arr.shape[0] == nrows
        rep = 1 if arr.ndim == 1 else arr.shape[1]
        if arr.dtype.kind == 'i':
            code, dt = 'J', '>i4'
        elif arr.dtype == np.float32:
            code, dt = 'E', '>f4'
        else:
            code, dt = 'D', '>f8'
        fields.append((name, dt, (rep,)) if rep > 1 else (name, dt))
        tforms.append('%d%s' % (rep, code))
    rec = np.empty(nrows, dtype=np.dtype(fields))
    for name, arr in columns:
        rec[name] = arr
    prim = _header([_card('SIMPLE', True), _card('BITPIX', 8), _card('NAXIS', 0), _card('EXTEND', True)])
    cards = [_card('XTENSION', 'BINTABLE'), _card('BITPIX', 8), _card('NAXIS', 2), _card('NAXIS1', rec.dtype.itemsize),
             _card('NAXIS2', nrows), _card('PCOUNT', 0), _card('GCOUNT', 1), _card('TFIELDS', len(columns))]
    for i, ((name, _), tf) in enumerate(zip(columns, tforms)):
        cards += [_card('TTYPE%d' % (i + 1), name), _card('TFORM%d' % (i + 1), tf)]
    cards.append(_card('EXTNAME', extname))
    for k, v in extra:
        cards.append(_card(k, v))
    data = rec.tobytes()
    tmp = fname + '.tmp%d' % os.getpid()
    with (gzip.open(tmp, 'wb') if fname.endswith('.gz') else open(tmp, 'wb')) as f:
        f.write(prim)
        f.write(_header(cards))
        f.write(data)
        f.write(b'\0' * ((-len(data)) % _BLOCK))
    os.replace(tmp, fname)


def _read_header(f):
    cards = {}
    while True:
        block = f.read(_BLOCK)
        if len(block) < _BLOCK:
            raise OSError('truncated FITS header')
        for i in range(0, _BLOCK, 80):
            c = block[i:i + 80].decode('ascii', 'replace')
            key = c[:8].strip()
            if key == 'END':
                return cards
            if c[8:10] == '= ':
                val = c[10:].split(' /')[0].strip()
                if val.startswith("'"):
                    val = val.strip("'").strip()
                elif val in ('T', 'F'):
                    val = val == 'T'
                else:
                    try:
                        val = int(val)
                    except ValueError:
                        try:
                            val = float(val.replace('D', 'E'))
                        except ValueError:
                            pass
                cards[key] = val


def read_bintable(fname, hdu=1):
    """Returns (dict name -> array, header dict) of binary-table extension number `hdu`."""
    with (gzip.open(fname, 'rb') if fname.endswith('.gz') else open(fname, 'rb')) as f:
        hdr = _read_header(f)  # primary
        nbytes = abs(hdr.get('BITPIX', 8)) // 8
        if hdr.get('NAXIS', 0) > 0:
            n = 1
            for i in range(hdr['NAXIS']):
                n *= hdr['NAXIS%d' % (i + 1)]
            f.read((n * nbytes + _BLOCK - 1) // _BLOCK * _BLOCK)
        for ih in range(1, hdu + 1):
            hdr = _read_header(f)
            size = hdr['NAXIS1'] * hdr['NAXIS2'] + hdr.get('PCOUNT', 0)
            if ih < hdu:
                f.read((size + _BLOCK - 1) // _BLOCK * _BLOCK)
        fields = []
        for i in range(1, hdr['TFIELDS'] + 1):
            tf = str(hdr['TFORM%d' % i]).strip()
            code = tf[-1]
            rep = int(tf[:-1]) if tf[:-1] else 1
            name = str(hdr.get('TTYPE%d' % i, 'col%d' % i)).strip()
            if code == 'A':  # character field of `rep` bytes
                fields.append((name, 'S%d' % rep))
                continue
            dt = {'J': '>i4', 'K': '>i8', 'E': '>f4', 'D': '>f8', 'I': '>i2', 'B': 'u1', 'L': 'S1'}[code]
            fields.append((name, dt, (rep,)) if rep > 1 else (name, dt))
        rec = np.frombuffer(f.read(hdr['NAXIS1'] * hdr['NAXIS2']), dtype=np.dtype(fields), count=hdr['NAXIS2'])
    return {n: np.ascontiguousarray(rec[n]) for n in rec.dtype.names}, hdr
