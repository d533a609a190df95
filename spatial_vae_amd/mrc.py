"""MRC / MRCS image stacks (the particle input format of train_particles.py; reference reader:
/root/reference/spatial_vae/mrc.py:108-140 `parse`, writer :178-218 `write`).

The 1024-byte header is described once as a numpy structured dtype (MRC2014 / IMOD field layout, little-endian)
instead of a struct format string; image data is returned as a MEMORY MAP by `read` (particle stacks run to tens of
GB: the training script slices what it needs and moves it to the device) or as a view of the caller's buffer by
`parse`.  Same return convention as the reference: (array, header, extended_header); nz == 1 drops the stack axis.
"""
from collections import namedtuple

import numpy as np

HEADER_BYTES = 1024
HEADER_DTYPE = np.dtype({
    "names": ["nx", "ny", "nz", "mode", "nxstart", "nystart", "nzstart", "mx", "my", "mz", "xlen", "ylen", "zlen",
              "alpha", "beta", "gamma", "mapc", "mapr", "maps", "amin", "amax", "amean", "ispg", "next", "creatid",
              "nint", "nreal", "imodStamp", "imodFlags", "idtype", "lens", "nd1", "nd2", "vd1", "vd2",
              "tilt_ox", "tilt_oy", "tilt_oz", "tilt_cx", "tilt_cy", "tilt_cz", "xorg", "yorg", "zorg", "cmap", "stamp",
              "rms", "nlabl", "labels"],
    "formats": ["<i4"] * 10 + ["<f4"] * 6 + ["<i4"] * 3 + ["<f4"] * 3 + ["<i4", "<i4", "<i2"] + ["<i2", "<i2"]
               + ["<i4", "<i4"] + ["<i2"] * 6 + ["<f4"] * 6 + ["<f4"] * 3 + ["S4", "S4", "<f4", "<i4", "S800"],
    "offsets": list(range(0, 40, 4)) + list(range(40, 64, 4)) + list(range(64, 76, 4)) + list(range(76, 88, 4))
               + [88, 92, 96] + [128, 130] + [152, 156] + list(range(160, 172, 2)) + list(range(172, 196, 4))
               + [196, 200, 204, 208, 212, 216, 220, 224],
    "itemsize": HEADER_BYTES,
})
MRCHeader = namedtuple("MRCHeader", HEADER_DTYPE.names)

# mode -> sample dtype (mode 3: complex from two int16, mode 16: RGB bytes; both come back with a trailing axis)
_MODES = {0: np.dtype("i1"), 1: np.dtype("<i2"), 2: np.dtype("<f4"), 3: np.dtype(("<i2", (2,))), 4: np.dtype("<c8"),
          6: np.dtype("<u2"), 16: np.dtype(("u1", (3,)))}


def _header(raw):
    if len(raw) < HEADER_BYTES:
        raise ValueError("MRC: file shorter than its %d-byte header" % HEADER_BYTES)
    rec = np.frombuffer(raw, dtype=HEADER_DTYPE, count=1)[0]
    vals = [rec[name].item() for name in HEADER_DTYPE.names]
    for i, name in enumerate(HEADER_DTYPE.names):                 # numpy strips trailing NULs from S fields; keep the width
        if HEADER_DTYPE[name].kind == "S":
            vals[i] = vals[i].ljust(HEADER_DTYPE[name].itemsize, b"\x00")
    h = MRCHeader(*vals)
    if h.mode not in _MODES:
        raise ValueError("MRC: unsupported mode %d" % h.mode)
    if h.nx < 0 or h.ny < 0 or h.nz < 0 or h.next < 0:
        raise ValueError("MRC: negative size in header (nx=%d ny=%d nz=%d next=%d)" % (h.nx, h.ny, h.nz, h.next))
    return h


def _shape(array, h):
    array = array.reshape((h.nz, h.ny, h.nx) + array.shape[1:])
    return array[0] if h.nz == 1 else array


def parse(content):
    """bytes-like whole file -> (array, header, extended_header); the array is a read-only view of `content`."""
    h = _header(content[:HEADER_BYTES])
    start = HEADER_BYTES + h.next
    dt = _MODES[h.mode]
    count = h.nx * h.ny * h.nz
    if len(content) < start + count * dt.itemsize:
        raise ValueError("MRC: %d bytes of image data expected, %d present" % (count * dt.itemsize, len(content) - start))
    array = np.frombuffer(content, dtype=dt, count=count, offset=start)
    return _shape(array, h), h, bytes(content[HEADER_BYTES:start])


def read(path):
    """File -> (memory-mapped array, header, extended_header) without loading the stack into host memory."""
    with open(path, "rb") as f:
        raw = f.read(HEADER_BYTES)
        h = _header(raw)
        ext = f.read(h.next)
    dt = _MODES[h.mode]
    count = h.nx * h.ny * h.nz
    if count == 0:
        return _shape(np.zeros((0,) + dt.shape, dt.base), h), h, ext
    array = np.memmap(path, dtype=dt, mode="r", offset=HEADER_BYTES + h.next, shape=(count,))
    return _shape(array, h), h, ext


def mode_of(dtype):
    dtype = np.dtype(dtype)
    for mode, dt in _MODES.items():
        if dt == dtype or (dt.subdtype is None and dt.newbyteorder("=") == dtype.newbyteorder("=")):
            return mode
    raise TypeError("MRC incompatible dtype: %s" % dtype)          # the reference raises a str here (a py3 TypeError)


def write(f, array, extended_header=b"", ax=1, ay=1, az=1, alpha=0, beta=0, gamma=0):
    """Write a (nz, ny, nx) stack (or one (ny, nx) image) to the open binary file f with a fresh header: cell
    (ax, ay, az, alpha, beta, gamma), axis order 1 2 3, density statistics of the data."""
    array = np.asarray(array)
    stack = array[None] if array.ndim == 2 else array
    mode = mode_of(stack.dtype)
    rec = np.zeros(1, HEADER_DTYPE)
    rec["nx"], rec["ny"], rec["nz"] = stack.shape[2], stack.shape[1], stack.shape[0]
    rec["mode"] = mode
    rec["mx"] = rec["my"] = rec["mz"] = 1
    rec["xlen"], rec["ylen"], rec["zlen"] = ax, ay, az
    rec["alpha"], rec["beta"], rec["gamma"] = alpha, beta, gamma
    rec["mapc"], rec["mapr"], rec["maps"] = 1, 2, 3
    if stack.size:
        rec["amin"], rec["amax"], rec["amean"], rec["rms"] = stack.min(), stack.max(), stack.mean(), stack.std()
    rec["next"] = len(extended_header)
    f.write(rec.tobytes())
    f.write(extended_header)
    f.write(np.ascontiguousarray(stack, dtype=_MODES[mode].base.newbyteorder("<")).tobytes())
