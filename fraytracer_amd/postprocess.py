"""Post-processing of the rendered FColor[,] — OUTSIDE the hot path, host side.

Image.toColors (src/FrayTracer/Image.fs:37-50) + FColor.gammaInverse / toColor (FColor.fs:43-55) and the
24-bpp BMP of Image.toBitmap / saveBitmap (Image.fs:61-90).  The reference draws the dithering noise from
ONE System.Random shared by a parallel map (Image.fs:46-49), so its low bit is nondeterministic; here the
draws are made sequentially in [x, y], R, G, B order — comparable to the reference only to ±1 LSB.
"""
import struct

import numpy as np

F = np.float32


def toColors(gamma, rng, image):
    """Image.toColors gamma rng image -> uint8 [X, Y, 3] (R, G, B)."""
    image = np.asarray(image, F)
    gammaInv = F(1.0) / F(gamma)
    mx = max(F(image.max()), F(0.01))                                   # Image.fs:40-43
    c = np.power(image / mx, gammaInv, dtype=F)                         # FColor.fs:50-55 MathF.Pow
    X, Y, _ = c.shape
    noise = np.array([rng.range_01() for _ in range(X * Y * 3)], F).reshape(X, Y, 3) if rng is not None else F(0.5)
    v = c * F(254.5) + noise                                            # FColor.fs:45-47
    return np.minimum(np.rint(v), 255).astype(np.uint8)                 # MathF.Round = half-to-even; `min 255`


def toBitmapRows(colors):
    """Image.toBitmap (Image.fs:61-86): after the index arithmetic and Array.rev the bitmap pixel at
    (column c, row r counted from the top) is image[X-1-c, r].  Returns uint8 [rows=Y, cols=X, 3] as B, G, R."""
    colors = np.asarray(colors, np.uint8)
    rows = np.transpose(colors[::-1, :, :], (1, 0, 2))                  # [r, c] = colors[X-1-c, r]
    return rows[:, :, ::-1]


def saveBitmap(path, colors):
    """Image.saveBitmap: uncompressed 24-bpp BMP (bottom-up rows, 4-byte row padding)."""
    rows = toBitmapRows(colors)
    h, w, _ = rows.shape
    pad = (-3 * w) % 4
    body = b"".join(rows[r].tobytes() + b"\0" * pad for r in range(h - 1, -1, -1))
    header = struct.pack("<2sIHHI", b"BM", 54 + len(body), 0, 0, 54)
    info = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(body), 3780, 3780, 0, 0)
    with open(path, "wb") as f:
        f.write(header + info + body)
