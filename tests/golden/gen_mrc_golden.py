"""Write the MRC fixtures with the REFERENCE's own writer/parser (/root/reference/spatial_vae/mrc.py) -- run in the
build container only:  python tests/golden/gen_mrc_golden.py
Outputs: mrc_ref_*.mrc(s) = files produced by the reference's `write`; mrc_golden.npz = the arrays and header fields its
`parse` returns for them (and for one file written by OUR writer, which the reference parser must read back)."""
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_mrc", "/root/reference/spatial_vae/mrc.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

from spatial_vae_amd import mrc as ours  # noqa: E402


def main():
    rs = np.random.RandomState(11)
    out = {}
    stacks = {"f32_stack": rs.normal(size=(3, 5, 4)).astype(np.float32),
              "f32_single": rs.normal(size=(1, 6, 6)).astype(np.float32)}
    for name, arr in stacks.items():
        path = os.path.join(HERE, "mrc_ref_%s.mrcs" % name)
        with open(path, "wb") as f:
            ref.write(f, arr, extended_header=b"EXT0" * 5 if name == "f32_stack" else b"", ax=4.0, ay=5.0, az=3.0)
        got, hdr, ext = ref.parse(open(path, "rb").read())
        out[name] = np.array(got)
        out[name + ".header"] = np.array([hdr.nx, hdr.ny, hdr.nz, hdr.mode, hdr.next, hdr.mapc, hdr.mapr, hdr.maps], np.int64)
        out[name + ".stats"] = np.array([hdr.amin, hdr.amax, hdr.amean, hdr.rms, hdr.xlen, hdr.ylen, hdr.zlen], np.float64)
    # integer modes: the reference writer always stamps mode 2, so build those files from its header maker
    for name, arr in {"i16": rs.randint(-300, 300, size=(2, 3, 7)).astype(np.int16),
                      "u16": rs.randint(0, 60000, size=(2, 4, 4)).astype(np.uint16),
                      "i8": rs.randint(-100, 100, size=(4, 2, 3)).astype(np.int8)}.items():
        hdr = ref.make_header(arr.shape, (1, 1, 1), (90, 90, 90), dtype=arr.dtype)
        path = os.path.join(HERE, "mrc_ref_%s.mrc" % name)
        with open(path, "wb") as f:
            ref.write(f, arr, header=hdr)
        got, h2, _ = ref.parse(open(path, "rb").read())
        assert np.array_equal(got, arr)
        out[name] = arr
    # our writer -> reference parser
    buf = io.BytesIO()
    ours.write(buf, stacks["f32_stack"], extended_header=b"EXT0" * 5, ax=4.0, ay=5.0, az=3.0)
    got, hdr, ext = ref.parse(buf.getvalue())
    assert np.array_equal(got, stacks["f32_stack"]) and ext == b"EXT0" * 5 and hdr.mode == 2
    assert buf.getvalue() == open(os.path.join(HERE, "mrc_ref_f32_stack.mrcs"), "rb").read(), "writers differ byte-wise"
    np.savez_compressed(os.path.join(HERE, "mrc_golden.npz"), **out)
    print("mrc fixtures written:", sorted(out))


if __name__ == "__main__":
    main()
