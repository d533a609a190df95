"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's CTF filter bank,
/root/reference/spatial_vae/ctf.py:7-24 (closed-form 2-D CTF) and :33-56 (evaluate it on the FFT frequency grid,
ifft2 + fftshift, real part, negated).  The product builds its filters on the device (svae_ctf_filter); this file is
what tests/ compare that kernel with, and is itself pinned against filters the reference's own ctf.py wrote
(tests/golden/ctf_golden.npz, tests/test_cli_cpu.py).  Only tests/ may import it.  `params` is the column dict of
spatial_vae_amd.ctf.parse_ctf."""
import numpy as np

COLUMNS = ("defocus", "cs", "voltage", "apix", "bfactor", "ampcont", "dfdiff", "dfang")


def ctf_2d(freqs, dfu, dfv, dfang, volt_kv, cs_mm, w, bfactor=None):
    """Contrast transfer function at spatial frequencies freqs (K, 2) [1/Angstrom]."""
    volt = volt_kv * 1000.0
    cs = cs_mm * 1e7
    lam = 12.2639 / np.sqrt(volt + 0.97845e-6 * volt ** 2)          # electron wavelength, Angstrom
    fx, fy = freqs[:, 0], freqs[:, 1]
    s2 = fx ** 2 + fy ** 2
    df = 0.5 * (dfu + dfv + (dfu - dfv) * np.cos(2 * (np.arctan2(fy, fx) - dfang)))
    gamma = 2 * np.pi * (-0.5 * df * lam * s2 + 0.25 * cs * lam ** 3 * s2 ** 2)
    out = np.sqrt(1 - w ** 2) * np.sin(gamma) - w * np.cos(gamma)
    if bfactor is not None:
        out = out * np.exp(-bfactor / 4 * s2)
    return out.astype(freqs.dtype)


def ctf_filter(params, n, m, scale=1):
    """(P, n, m) real-space filters, one per particle."""
    ty, tx = np.meshgrid(np.fft.fftfreq(n), np.fft.fftfreq(m), indexing="ij")
    freqs = np.stack([ty.ravel(), tx.ravel()], 1)
    count = len(params["defocus"])
    out = np.zeros((count, n, m), dtype=np.float32)
    for i in range(count):
        apix = params["apix"][i] * scale
        c = ctf_2d(freqs / apix, params["defocus"][i] * 10000, params["defocus"][i] * 10000,
                   2 * np.pi * params["dfang"][i] / 360, params["voltage"][i], params["cs"][i],
                   params["ampcont"][i] / 100, params["bfactor"][i]).reshape(n, m)
        out[i] = -np.fft.fftshift(np.fft.ifft2(c)).real
    return out
