"""spatial_vae_amd/mrc.py against files written by the reference's own MRC writer and the arrays its parser returned
for them (tests/golden/mrc_ref_*.mrc[s], mrc_golden.npz; generator: tests/golden/gen_mrc_golden.py).  CPU only."""
import io
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from spatial_vae_amd import mrc


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(GOLDEN_DIR, "mrc_golden.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.mark.parametrize("name", ["f32_stack", "f32_single"])
def test_reads_reference_written_float_stacks(name, gold):
    path = os.path.join(GOLDEN_DIR, "mrc_ref_%s.mrcs" % name)
    arr, hdr, ext = mrc.read(path)
    assert isinstance(arr, np.memmap) or arr.base is not None              # mapped, not loaded
    assert np.array_equal(np.asarray(arr), gold[name]) and arr.dtype == np.float32
    assert [hdr.nx, hdr.ny, hdr.nz, hdr.mode, hdr.next, hdr.mapc, hdr.mapr, hdr.maps] == list(gold[name + ".header"])
    assert np.allclose([hdr.amin, hdr.amax, hdr.amean, hdr.rms, hdr.xlen, hdr.ylen, hdr.zlen], gold[name + ".stats"], rtol=0, atol=0)
    assert ext == (b"EXT0" * 5 if name == "f32_stack" else b"")
    arr2, hdr2, ext2 = mrc.parse(open(path, "rb").read())                   # the bytes entry the reference exposes
    assert np.array_equal(arr2, gold[name]) and hdr2 == hdr and ext2 == ext
    if name == "f32_single":
        assert arr.shape == (6, 6)                                          # nz == 1 drops the stack axis (mrc.py:135-136)


@pytest.mark.parametrize("name,dtype", [("i16", np.int16), ("u16", np.uint16), ("i8", np.int8)])
def test_reads_integer_modes(name, dtype, gold):
    arr, hdr, _ = mrc.read(os.path.join(GOLDEN_DIR, "mrc_ref_%s.mrc" % name))
    assert arr.dtype == dtype and np.array_equal(np.asarray(arr), gold[name])
    assert hdr.mode == mrc.mode_of(dtype)


def test_writer_is_byte_identical_to_the_reference_writer(gold):
    buf = io.BytesIO()
    mrc.write(buf, gold["f32_stack"], extended_header=b"EXT0" * 5, ax=4.0, ay=5.0, az=3.0)
    assert buf.getvalue() == open(os.path.join(GOLDEN_DIR, "mrc_ref_f32_stack.mrcs"), "rb").read()


def test_round_trip_and_errors(tmp_path):
    rs = np.random.RandomState(0)
    a = rs.normal(size=(7, 12, 10)).astype(np.float32)
    p = tmp_path / "stack.mrcs"
    with open(p, "wb") as f:
        mrc.write(f, a)
    b, hdr, _ = mrc.read(str(p))
    assert np.array_equal(np.asarray(b), a) and (hdr.nx, hdr.ny, hdr.nz) == (10, 12, 7)
    raw = bytearray(open(p, "rb").read())
    with pytest.raises(ValueError, match="image data expected"):
        mrc.parse(bytes(raw[:-8]))                                          # truncated data
    with pytest.raises(ValueError, match="shorter than"):
        mrc.parse(bytes(raw[:100]))
    raw[12:16] = (99).to_bytes(4, "little")
    with pytest.raises(ValueError, match="unsupported mode"):
        mrc.parse(bytes(raw))
    with pytest.raises(TypeError):
        mrc.mode_of(np.float64)
    assert mrc.mode_of(np.dtype(("u1", (3,)))) == 16 and mrc.mode_of(np.complex64) == 4
