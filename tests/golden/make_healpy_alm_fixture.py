"""Builds tests/golden/healpy_alm_lmax3.fits: the file healpy.write_alm produces for a small alm array, assembled byte by byte
from the documented format -- independently of plancklens_amd/fitsio.py, which the fixture pins (the
reference reads and writes these files at plancklens/qest.py:17,201, filt/filt_simple.py:97-99).

healpy.write_alm(filename, alm) with a complex128 array writes a primary HDU without data and ONE binary-table extension with
three scalar columns, one row per stored (l, m), m-major as the alm array itself:
    index  TFORM 'J' (32-bit big-endian integer)  = l*l + l + m + 1      (TUNIT 'l*l+l+m+1')
    real   TFORM 'D' (64-bit big-endian IEEE)     = Re a_lm
    imag   TFORM 'D'                              = Im a_lm
FITS standard: 80-character cards, header and data units padded to multiples of 2880 bytes (header with blanks, data with
zeros), string values left-justified in quotes and padded to at least 8 characters, numbers right-justified to column 30.

    python tests/golden/make_healpy_alm_fixture.py
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
LMAX = 3


def alm_values():
    """a_lm = (l + m / 10) + i (m - l / 100) for m > 0, real for m = 0; m-major order"""
    rows = []
    for m in range(LMAX + 1):
        for l in range(m, LMAX + 1):
            rows.append((l * l + l + m + 1, l + m / 10., (m - l / 100.) if m > 0 else 0.))
    return rows


def card(key, value, comment=''):
    if isinstance(value, bool):
        v = ('T' if value else 'F').rjust(20)
    elif isinstance(value, int):
        v = str(value).rjust(20)
    else:
        v = "'%s'" % str(value).ljust(8)
        v = v.ljust(20)
    s = '%-8s= %s' % (key, v)
    if comment:
        s += ' / ' + comment
    assert len(s) <= 80
    return s.ljust(80)


def unit(cards):
    txt = ''.join(cards) + 'END'.ljust(80)
    return (txt + ' ' * (-len(txt) % 2880)).encode('ascii')


def build():
    rows = alm_values()
    primary = unit([card('SIMPLE', True, 'conforms to FITS standard'), card('BITPIX', 8, 'array data type'),
                    card('NAXIS', 0, 'number of array dimensions'), card('EXTEND', True)])
    ext = unit([card('XTENSION', 'BINTABLE', 'binary table extension'), card('BITPIX', 8, 'array data type'),
                card('NAXIS', 2, 'number of array dimensions'), card('NAXIS1', 20, 'length of dimension 1'),
                card('NAXIS2', len(rows), 'length of dimension 2'), card('PCOUNT', 0, 'number of group parameters'),
                card('GCOUNT', 1, 'number of groups'), card('TFIELDS', 3, 'number of table fields'),
                card('TTYPE1', 'index'), card('TFORM1', 'J'), card('TUNIT1', 'l*l+l+m+1'),
                card('TTYPE2', 'real'), card('TFORM2', 'D'), card('TUNIT2', 'unknown'),
                card('TTYPE3', 'imag'), card('TFORM3', 'D'), card('TUNIT3', 'unknown')])
    data = b''.join(struct.pack('>idd', i, re, im) for i, re, im in rows)
    data += b'\0' * (-len(data) % 2880)
    return primary + ext + data


if __name__ == '__main__':
    out = os.path.join(HERE, 'healpy_alm_lmax3.fits')
    open(out, 'wb').write(build())
    print('wrote', out, os.path.getsize(out), 'bytes')
