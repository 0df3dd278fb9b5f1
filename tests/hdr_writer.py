"""Radiance RGBE (.hdr) files for the tests: shared-exponent pixels, written flat or with the per-row run-length
coding, under either identifier line."""
import numpy as np


def to_rgbe(rgb):
    """(h, w, 3) floats -> (h, w, 4) bytes: mantissas and the shared exponent, black as four zeros."""
    rgb = np.asarray(rgb, np.float64)
    top = rgb.max(axis=-1)
    mantissa, exponent = np.frexp(top)
    scale = np.where(top > 1e-32, mantissa * 256.0 / np.maximum(top, 1e-300), 0.0)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(top > 1e-32, exponent + 128, 0).astype(np.uint8)
    return out


def _run_length(channel):
    out, i, n = bytearray(), 0, len(channel)
    while i < n:
        run = 1
        while i + run < n and run < 127 and channel[i + run] == channel[i]:
            run += 1
        if run >= 3:
            out += bytes((128 + run, channel[i]))
            i += run
            continue
        start = i
        while i < n and i - start < 128:
            if i + 2 < n and channel[i] == channel[i + 1] == channel[i + 2]:
                break
            i += 1
        out += bytes((i - start,)) + bytes(channel[start:i])
    return bytes(out)


def write_hdr(filename, rgbe, run_length=True, identifier="#?RADIANCE", extra_header=("EXPOSURE=1.0",)):
    h, w = rgbe.shape[:2]
    body = bytearray()
    for row in rgbe:
        if run_length and 8 <= w < 32768:
            body += bytes((2, 2, w >> 8, w & 255))
            for k in range(4):
                body += _run_length(row[:, k].tolist())
        else:
            body += row.tobytes()
    header = "\n".join((identifier,) + tuple(extra_header) + ("FORMAT=32-bit_rle_rgbe", "", f"-Y {h} +X {w}")) + "\n"
    with open(filename, "wb") as f:
        f.write(header.encode() + bytes(body))
