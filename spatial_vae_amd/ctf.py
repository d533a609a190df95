"""CTF parameter tables and the host form of the filter bank -- /root/reference/spatial_vae/ctf.py:7-56 (parse the
8-column parameter table, evaluate the closed-form 2-D CTF on the FFT frequency grid, bring it to real space with
ifft2 + fftshift, negate).  The training script builds its filters on the device (ops.ctf_filter ->
svae_ctf_filter); `ctf_filter` below is the same computation with numpy for hosts without a GPU (table
inspection, tests).  Filter APPLICATION is svae_gaussian_loglik's job.  numpy only (the reference uses pandas
just to read the whitespace table)."""
import numpy as np

COLUMNS = ("defocus", "cs", "voltage", "apix", "bfactor", "ampcont", "dfdiff", "dfang")


def parse_ctf(path):
    """Whitespace table, one particle per row -> dict of column arrays."""
    tab = np.loadtxt(path, ndmin=2)
    if tab.shape[1] != len(COLUMNS):
        raise ValueError("expected %d CTF columns, got %d" % (len(COLUMNS), tab.shape[1]))
    return {name: tab[:, i] for i, name in enumerate(COLUMNS)}


def ctf_table(params):
    """Column dict -> (P, 8) float64 table in COLUMNS order (the `params` operand of svae_ctf_filter)."""
    return np.stack([np.asarray(params[name], np.float64) for name in COLUMNS], 1)


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
