"""A small PNG writer for tests: any colour type and bit depth, Adam7 interlacing, a chosen row filter, optional
PLTE / tRNS chunks - the cases an off-the-shelf encoder does not produce."""
import struct
import zlib

import numpy as np

CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)


def _pack_rows(samples, depth):
    """samples: (h, w, c) unsigned -> list of packed rows (bytes)."""
    h, w, c = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1).astype(np.uint32)
        if depth == 16:
            rows.append(flat.astype(">u2").tobytes())
        elif depth == 8:
            rows.append(flat.astype(np.uint8).tobytes())
        else:
            per = 8 // depth
            padded = np.concatenate([flat, np.zeros((-len(flat)) % per, np.uint32)]).reshape(-1, per)
            shifts = np.arange(per - 1, -1, -1) * depth
            rows.append((padded << shifts).sum(axis=1).astype(np.uint8).tobytes())
    return rows


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp, choose):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        f = choose(y)
        line = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            line[i] = (v - pred) & 255
        out.append(f)
        out += line
        prev = row
    return bytes(out)


def write_png(path, samples, colour_type, depth, interlace=False, row_filter=None, palette=None, trns=None, idat_split=0):
    """samples: (h, w, channels) array of unsigned sample values below 2**depth (palette indices for type 3)."""
    samples = np.asarray(samples)
    h, w, c = samples.shape
    assert c == CHANNELS[colour_type]
    bpp = max(1, c * depth // 8)
    choose = (lambda y: row_filter) if row_filter is not None else (lambda y: y % 5)
    raw = b""
    if interlace:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(_pack_rows(sub, depth), bpp, choose)
    else:
        raw = _filter_rows(_pack_rows(samples, depth), bpp, choose)
    data = zlib.compress(raw, 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour_type, 0, 0, 1 if interlace else 0))
    out += _chunk(b"gAMA", struct.pack(">I", 45455))           # ancillary chunks are skipped
    if palette is not None:
        out += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    if idat_split:
        for i in range(0, len(data), idat_split):
            out += _chunk(b"IDAT", data[i:i + idat_split])
    else:
        out += _chunk(b"IDAT", data)
    out += _chunk(b"tEXt", b"Comment\0test") + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)
