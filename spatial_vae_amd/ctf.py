"""CTF parameter tables -- /root/reference/spatial_vae/ctf.py:26-30 (parse the 8-column parameter table).  The
filters themselves are built on the device (ops.ctf_filter -> svae_ctf_filter, replacing ctf.py:7-24, 33-56) and
applied by svae_gaussian_loglik; the numpy restatement of the filter computation that the tests check the kernel
against lives with the other checkers (ctf_oracle.py in the test oracle directory).  numpy only (the reference uses pandas just to read
the whitespace table)."""
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
