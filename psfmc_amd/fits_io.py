"""
Minimal FITS image I/O (astropy is not available next to the ROCm stack).

Covers what the hot path's setup needs (psfMC/utils.py:61-63, 87, 111-112 call
`astropy.io.fits.getdata/getheader`): read the first HDU that holds an image
(primary or IMAGE extension, BITPIX 8/16/32/64/-32/-64, optional
BSCALE/BZERO), gzip-compressed files included, and write a single-HDU image.
Float data keep their on-disk precision (float32 stays float32), which matters
because the reference's raw-model dtype follows the FITS dtype (models.py:249).
"""
import gzip
from collections import OrderedDict

import numpy as np

BLOCK = 2880
_BITPIX = {8: 'u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}


def _open(path):
    with open(path, 'rb') as f:
        magic = f.read(2)
    return gzip.open(path, 'rb') if magic == b'\x1f\x8b' else open(path, 'rb')


def _parse_value(text):
    text = text.strip()
    if not text:
        return None
    if text.startswith("'"):
        end = text.find("'", 1)
        while end != -1 and text[end:end + 2] == "''":
            end = text.find("'", end + 2)
        return text[1:end if end != -1 else None].replace("''", "'").rstrip()
    text = text.split('/')[0].strip()
    if text in ('T', 'F'):
        return text == 'T'
    try:
        return int(text)
    except ValueError:
        try:
            return float(text.replace('D', 'E'))
        except ValueError:
            return text


def _read_header(f):
    """Returns (OrderedDict, found_end).  Tolerates a missing END card
    (`ignore_missing_end=True` in the reference's calls)."""
    hdr = OrderedDict()
    while True:
        block = f.read(BLOCK)
        if len(block) < BLOCK:
            return hdr, False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode('ascii', 'replace')
            key = card[:8].strip()
            if key == 'END':
                return hdr, True
            if card[8:10] == '= ' and key:
                hdr[key] = _parse_value(card[10:])
            elif key in ('COMMENT', 'HISTORY'):
                hdr.setdefault(key, [])
                hdr[key].append(card[8:].rstrip())


def _data_shape(hdr):
    naxis = int(hdr.get('NAXIS', 0))
    return tuple(int(hdr['NAXIS%d' % (k + 1)]) for k in reversed(range(naxis)))


def read_image(path, with_header=False):
    """First image found in the file, as a native-endian array."""
    if not isinstance(path, str):
        raise IOError('not a file name: %r' % (path,))
    with _open(path) as f:
        first = True
        while True:
            hdr, _ = _read_header(f)
            if not hdr:
                raise IOError('no image data in FITS file %s' % path)
            if first and hdr.get('SIMPLE') is not True:
                raise IOError('%s is not a FITS file' % path)
            first = False
            shape = _data_shape(hdr)
            count = int(np.prod(shape)) if shape else 0
            bitpix = int(hdr.get('BITPIX', 8))
            nbytes = count * abs(bitpix) // 8
            nbytes += int(hdr.get('PCOUNT', 0))
            is_image = hdr.get('XTENSION', 'IMAGE').strip() == 'IMAGE'
            if count and is_image:
                raw = f.read(count * abs(bitpix) // 8)
                data = np.frombuffer(raw, dtype=_BITPIX[bitpix], count=count)
                data = data.reshape(shape)
                data = data.astype(data.dtype.newbyteorder('='))
                bscale = hdr.get('BSCALE', 1)
                bzero = hdr.get('BZERO', 0)
                if bscale != 1 or bzero != 0:
                    if bitpix == 16 and bscale == 1 and bzero == 32768:
                        data = (data.astype(np.int32) + 32768).astype(np.uint16)
                    else:
                        data = data * np.float64(bscale) + np.float64(bzero)
                return (data, hdr) if with_header else data
            f.read(-(-nbytes // BLOCK) * BLOCK)      # skip this HDU's data


def read_header(path):
    with _open(path) as f:
        hdr, _ = _read_header(f)
    return hdr


def _card(key, value, comment=''):
    if isinstance(value, bool):
        val = 'T' if value else 'F'
        text = '%-8s= %20s' % (key, val)
    elif isinstance(value, (int, np.integer)):
        text = '%-8s= %20d' % (key, value)
    elif isinstance(value, (float, np.floating)):
        text = '%-8s= %20s' % (key, ('%.15G' % value))
    else:
        text = "%-8s= '%-8s'" % (key, str(value).replace("'", "''")[:67])
    if comment:
        text = (text + ' / ' + comment)
    return text[:80].ljust(80)


def write_image(path, data, header=None):
    """Single-HDU FITS image.  `header`: mapping of extra cards (values
    bool/int/float/str)."""
    data = np.asarray(data)
    kinds = {'float32': -32, 'float64': -64, 'uint8': 8, 'int16': 16,
             'int32': 32, 'int64': 64, 'bool': 8}
    if data.dtype.name not in kinds:
        data = data.astype(np.float64)
    bitpix = kinds[data.dtype.name]
    cards = [_card('SIMPLE', True, 'conforms to FITS standard'),
             _card('BITPIX', bitpix), _card('NAXIS', data.ndim)]
    for k, n in enumerate(reversed(data.shape)):
        cards.append(_card('NAXIS%d' % (k + 1), int(n)))
    reserved = {'SIMPLE', 'BITPIX', 'NAXIS', 'END', 'EXTEND', 'BSCALE', 'BZERO'}
    for key, val in (header or {}).items():
        key = str(key).upper()[:8]
        if key in reserved or key.startswith('NAXIS') or isinstance(val, list):
            continue
        if val is None:
            continue
        cards.append(_card(key, val))
    cards.append('END'.ljust(80))
    head = ''.join(cards).encode('ascii')
    head += b' ' * (-len(head) % BLOCK)
    body = data.astype(_BITPIX[bitpix]).tobytes()
    body += b'\0' * (-len(body) % BLOCK)
    with open(path, 'wb') as f:
        f.write(head)
        f.write(body)


# --------------------------------------------------------------------------
# binary tables (trace database, psfMC/database.py:6-56 uses astropy.table)
# --------------------------------------------------------------------------
_TFORM = {'f8': ('D', '>f8'), 'i8': ('K', '>i8'), 'f4': ('E', '>f4'), 'i4': ('J', '>i4'),
          'b1': ('L', 'u1')}
_TFORM_READ = {'D': '>f8', 'K': '>i8', 'E': '>f4', 'J': '>i4', 'I': '>i2', 'B': 'u1', 'L': 'u1'}


def write_table(path, columns, meta=None):
    """One BINTABLE extension.  `columns`: ordered mapping name -> array [nrows] or
    [nrows, k]; `meta`: mapping key -> value or (value, comment)."""
    names = list(columns)
    arrays = [np.asarray(columns[n]) for n in names]
    nrows = len(arrays[0]) if arrays else 0
    fields, tforms, tdims = [], [], []
    for n, a in zip(names, arrays):
        if len(a) != nrows:
            raise ValueError('column {} has {} rows, expected {}'.format(n, len(a), nrows))
        kind = a.dtype.kind + str(a.dtype.itemsize)
        if kind not in _TFORM:
            a = a.astype(np.float64)
            kind = 'f8'
        width = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
        code, be = _TFORM[kind]
        fields.append((a.reshape(nrows, width), be, width))
        tforms.append(('%d%s' % (width, code)) if width > 1 else code)
        # astropy.table records the per-row shape of every multidimensional column, [n, 1]
        # ones included (the reference's scalar parameters are such columns, database.py:24)
        tdims.append('(%s)' % ','.join(str(d) for d in reversed(a.shape[1:])) if a.ndim > 1 else None)
    rec = np.dtype([('c%d' % i, be, (w,)) for i, (_, be, w) in enumerate(fields)])
    table = np.zeros(nrows, dtype=rec)
    for i, (a, be, w) in enumerate(fields):
        table['c%d' % i] = a
    primary = [_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', 8),
               _card('NAXIS', 0), _card('EXTEND', True), 'END'.ljust(80)]
    head0 = ''.join(primary).encode('ascii')
    head0 += b' ' * (-len(head0) % BLOCK)
    cards = [_card('XTENSION', 'BINTABLE', 'binary table extension'), _card('BITPIX', 8),
             _card('NAXIS', 2), _card('NAXIS1', rec.itemsize), _card('NAXIS2', nrows),
             _card('PCOUNT', 0), _card('GCOUNT', 1), _card('TFIELDS', len(names))]
    for i, (n, tf, td) in enumerate(zip(names, tforms, tdims)):
        cards.append(_card('TTYPE%d' % (i + 1), n))
        cards.append(_card('TFORM%d' % (i + 1), tf))
        if td is not None:
            cards.append(_card('TDIM%d' % (i + 1), td))
    for key, val in (meta or {}).items():
        comment = ''
        if isinstance(val, tuple):
            val, comment = val
        if isinstance(val, (np.bool_,)):
            val = bool(val)
        cards.append(_card(str(key).upper()[:8], val, comment))
    cards.append('END'.ljust(80))
    head1 = ''.join(cards).encode('ascii')
    head1 += b' ' * (-len(head1) % BLOCK)
    body = table.tobytes()
    body += b'\0' * (-len(body) % BLOCK)
    with open(path, 'wb') as f:
        f.write(head0 + head1 + body)


def read_table(path):
    """First BINTABLE extension -> (OrderedDict name -> native array, header)."""
    with _open(path) as f:
        hdr, _ = _read_header(f)
        shape = _data_shape(hdr)
        skip = (int(np.prod(shape)) if shape else 0) * abs(int(hdr.get('BITPIX', 8))) // 8
        f.read(-(-skip // BLOCK) * BLOCK)
        while True:
            hdr, _ = _read_header(f)
            if not hdr:
                raise IOError('no binary table in {}'.format(path))
            n1, n2 = int(hdr.get('NAXIS1', 0)), int(hdr.get('NAXIS2', 0))
            if str(hdr.get('XTENSION', '')).strip() == 'BINTABLE':
                break
            f.read(-(-(n1 * n2 + int(hdr.get('PCOUNT', 0))) // BLOCK) * BLOCK)
        fields = []
        for i in range(int(hdr['TFIELDS'])):
            tform = str(hdr['TFORM%d' % (i + 1)]).strip()
            width = int(tform[:-1]) if tform[:-1] else 1
            tdim = str(hdr.get('TDIM%d' % (i + 1), '')).strip().strip('()')
            shape = tuple(int(d) for d in reversed(tdim.split(','))) if tdim else None
            fields.append((str(hdr['TTYPE%d' % (i + 1)]).strip(), _TFORM_READ[tform[-1]], width, shape))
        rec = np.dtype([('c%d' % i, be, (w,)) for i, (_, be, w, _) in enumerate(fields)])
        raw = np.frombuffer(f.read(n1 * n2), dtype=rec, count=n2)
    cols = OrderedDict()
    for i, (name, be, w, shape) in enumerate(fields):
        a = raw['c%d' % i]
        a = a.astype(a.dtype.newbyteorder('='))
        if shape is not None and int(np.prod(shape)) == w:      # TDIM: the per-row shape
            cols[name] = a.reshape((len(a),) + shape)
        else:
            cols[name] = a[:, 0] if w == 1 else a
    return cols, hdr
