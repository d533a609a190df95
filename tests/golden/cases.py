"""Golden-vector case definitions and the deterministic input builder.

Shared by ``gen_golden.py`` (which runs the *reference* on these inputs, in the
build container only) and by the parity tests (which run the oracle and the HIP
path on the same inputs).  Everything here is generated from numpy
``RandomState`` streams so the inputs never have to be stored: fixtures hold
only the reference's outputs.

Parameter names follow the reference modules so a state dict built here loads
into either implementation:
  SpatialGenerator  (/root/reference/spatial_vae/models.py:57-88)
  InferenceNetwork  (/root/reference/spatial_vae/models.py:24-44)
"""
from collections import OrderedDict

import numpy as np

ACTS = ("tanh", "leakyrelu", "relu", "sigmoid")


def _case(name, script="mnist", n=7, m=7, B=4, z_dim=2, H=20, L=2, act="tanh",
          rotate=True, translate=True, dx_scale=0.1, theta_prior=np.pi / 4,
          z_scale=1.0, n_out=1, softplus=False, resid=False, expand_coords=False,
          bilinear=False, q_hidden=24, q_layers=1, mask=False, ctf=False,
          wscale=1.0, seed=0, store="full", augment=False):
    return dict(name=name, script=script, n=n, m=m, B=B, z_dim=z_dim, H=H, L=L,
                act=act, rotate=rotate, translate=translate, dx_scale=dx_scale,
                theta_prior=float(theta_prior), z_scale=z_scale, n_out=n_out,
                softplus=softplus, resid=resid, expand_coords=expand_coords,
                bilinear=bilinear, q_hidden=q_hidden, q_layers=q_layers,
                mask=mask, ctf=ctf, wscale=wscale, seed=seed, store=store, augment=augment)


# One row per variant of SURVEY.md section 8(a)/(c).  Sizes are tiny on purpose
# (N = 49 or 36 pixels straddles the 32-row tiles of the HIP path; H = 20 is not
# a multiple of 32).
CASES = [
    # train_mnist.py:24-90 (BCE, KL_theta with the mu^2 term)
    _case("mnist_rt", seed=1),
    _case("mnist_r", translate=False, seed=2),                       # BASELINE cfg 1 semantics
    _case("mnist_t", rotate=False, seed=3),
    _case("mnist_none", rotate=False, translate=False, seed=4),
    _case("mnist_L1", L=1, seed=5),
    _case("mnist_L3", L=3, H=33, seed=6),
    _case("mnist_L3_resid", L=3, resid=True, seed=7),
    _case("mnist_leaky", act="leakyrelu", seed=8),
    _case("mnist_relu", act="relu", seed=9),
    _case("mnist_sigmoid_act", act="sigmoid", seed=10),
    _case("mnist_expand", expand_coords=True, seed=11),
    _case("mnist_bilinear", bilinear=True, z_dim=3, seed=12),
    _case("mnist_bilinear_expand", bilinear=True, expand_coords=True, z_dim=3, seed=13),
    # softplus + BCE raises in the reference (softplus(sigmoid) > 1 fails binary_cross_entropy's
    # range check, train_mnist.py:81), so softplus is only pinned on the Gaussian path below.
    _case("mnist_z0", z_dim=0, seed=15),
    _case("mnist_saturated", wscale=40.0, seed=16),                  # sigmoid rounds to 1.0
    _case("mnist_dxscale", dx_scale=0.37, theta_prior=np.pi, seed=17),
    _case("mnist_wide", n=9, m=5, B=3, H=70, seed=18),               # non-square grid, 3 n-tiles
    # train_galaxy.py:27-128 (BCE over C channels, KL_theta without mu^2, z_scale)
    _case("galaxy_rgb", script="galaxy", n=6, m=6, n_out=3, L=3, z_dim=5, H=24,
          theta_prior=np.pi, z_scale=0.7, seed=20),
    _case("galaxy_rgb_relu", script="galaxy", n=6, m=6, n_out=3, L=2, z_dim=4, H=24,
          act="relu", theta_prior=np.pi, seed=21),
    # train_particles.py:22-148 (Gaussian log-likelihood; quirks A.3)
    _case("particles_gauss", script="particles", n=8, m=8, theta_prior=np.pi, seed=30),
    _case("particles_fit_noise", script="particles", n=8, m=8, n_out=2,
          theta_prior=np.pi, seed=31),
    _case("particles_mask", script="particles", n=8, m=8, mask=True,
          theta_prior=np.pi, seed=32),
    _case("particles_mask_fit_noise", script="particles", n=8, m=8, n_out=2, mask=True,
          theta_prior=np.pi, seed=33),
    _case("particles_ctf", script="particles", n=8, m=8, ctf=True, z_dim=3,
          theta_prior=np.pi, seed=34),
    _case("particles_ctf_mask", script="particles", n=8, m=8, ctf=True, mask=True,
          theta_prior=np.pi, z_scale=0.5, seed=35),
    _case("particles_resid_leaky", script="particles", n=8, m=8, L=3, resid=True,
          act="leakyrelu", theta_prior=np.pi, seed=36),
    _case("particles_softplus", script="particles", n=8, m=8, softplus=True,
          theta_prior=np.pi, seed=14),
    _case("particles_softplus_noise", script="particles", n=8, m=8, n_out=2, softplus=True,
          theta_prior=np.pi, seed=37),
    # --augment_rotation: Pillow bicubic rotation of the observed images by np.random angles before inference
    # (train_galaxy.py:41-54 through uint8; train_particles.py:31-43 as float32), offset added back to theta
    _case("galaxy_augment", script="galaxy", n=8, m=8, B=5, n_out=3, L=2, z_dim=3, H=24,
          theta_prior=np.pi, seed=50, augment=True),
    _case("particles_augment", script="particles", n=9, m=9, B=5, theta_prior=np.pi, seed=51, augment=True),
    # BASELINE.json widths (H=500 pads to 512; 28x28 pads to 25 tiles); outputs
    # stored as strided samples + norms to keep the fixture small.
    _case("mnist_h500", n=28, m=28, B=3, H=500, q_hidden=32, seed=40, store="sampled"),
    _case("particles_h500_noise", script="particles", n=40, m=40, B=2, H=500, n_out=2,
          q_hidden=32, theta_prior=np.pi, seed=41, store="sampled"),
]

CASES_BY_NAME = {c["name"]: c for c in CASES}

# cases whose fixture also holds the forward-only display outputs (minibatch_for_display, random_minibatch_generator)
DISPLAY_CASES = ("mnist_rt", "mnist_none", "mnist_z0", "galaxy_rgb", "galaxy_rgb_relu")


def inf_dim(case):
    return case["z_dim"] + (1 if case["rotate"] else 0) + (2 if case["translate"] else 0)


def coord_grid(n, m):
    """x_coord exactly as train_mnist.py:315-320 (float64 meshgrid -> float32)."""
    xgrid = np.linspace(-1, 1, m)
    ygrid = np.linspace(1, -1, n)
    x0, x1 = np.meshgrid(xgrid, ygrid)
    return np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)


def _linear(rs, n_out, n_in, wscale, bias=True):
    bound = 1.0 / np.sqrt(n_in)
    w = (rs.uniform(-bound, bound, size=(n_out, n_in)) * wscale).astype(np.float32)
    b = (rs.uniform(-bound, bound, size=(n_out,)) * wscale).astype(np.float32) if bias else None
    return w, b


def generator_state(case, rs):
    """Parameters of SpatialGenerator in attribute order (models.py:69-85)."""
    H, L, zd = case["H"], case["L"], case["z_dim"]
    in_dim = 5 if case["expand_coords"] else 2
    ws = case["wscale"]
    st = OrderedDict()
    w, b = _linear(rs, H, in_dim, ws)
    st["coord_linear.weight"], st["coord_linear.bias"] = w, b
    if zd > 0:
        st["latent_linear.weight"], _ = _linear(rs, H, zd, ws, bias=False)
    if zd > 0 and case["bilinear"]:
        bound = 1.0 / np.sqrt(in_dim)
        st["bilinear.weight"] = (rs.uniform(-bound, bound, size=(H, in_dim, zd)) * ws).astype(np.float32)
    idx = 1
    for _ in range(1, L):
        w, b = _linear(rs, H, H, ws)
        if case["resid"]:
            st["layers.%d.linear.weight" % idx], st["layers.%d.linear.bias" % idx] = w, b
            idx += 1
        else:
            st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = w, b
            idx += 2
    w, b = _linear(rs, case["n_out"], H, ws)
    st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = w, b
    return st


def inference_state(case, rs):
    """Parameters of InferenceNetwork (models.py:31-43)."""
    n_in = case["n"] * case["m"] * (case["n_out"] if case["script"] == "galaxy" else 1)
    Hq, Lq = case["q_hidden"], case["q_layers"]
    st = OrderedDict()
    w, b = _linear(rs, Hq, n_in, 1.0)
    st["layers.0.weight"], st["layers.0.bias"] = w, b
    idx = 2
    for _ in range(1, Lq):
        w, b = _linear(rs, Hq, Hq, 1.0)
        if case["resid"]:
            st["layers.%d.linear.weight" % idx], st["layers.%d.linear.bias" % idx] = w, b
            idx += 1
        else:
            st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = w, b
            idx += 2
    w, b = _linear(rs, 2 * inf_dim(case), Hq, 1.0)
    # keep log-std modest so z stays O(1)
    st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = (w * 0.5).astype(np.float32), b
    return st


def circular_mask(n, m):
    """train_particles.py:384-392."""
    radius = min(n, m) / 2
    y_grid, x_grid = np.ogrid[:n, :m]
    center = np.array([n / 2, m / 2])
    dist = np.sqrt((center[0] - y_grid) ** 2 + (center[1] - x_grid) ** 2)
    return (dist < radius).reshape(-1)


def build_inputs(case):
    """All inputs of one eval_minibatch call, as float32 numpy arrays."""
    rs = np.random.RandomState(1000 + case["seed"])
    n, m, B = case["n"], case["m"], case["B"]
    N = n * m
    out = dict(case=case)
    out["x_coord"] = coord_grid(n, m)
    out["p_state"] = generator_state(case, rs)
    out["q_state"] = inference_state(case, rs)
    if case["script"] == "mnist":
        u = rs.uniform(size=(B, N))
        keep = rs.uniform(size=(B, N)) > 0.6
        y = np.floor(u * keep * 255.0) / 255.0                      # uint8-quantised /255, sparse
    elif case["script"] == "galaxy":
        y = np.floor(rs.uniform(size=(B, N, case["n_out"])) * 255.0) / 255.0
    else:
        y = rs.normal(size=(B, N))                                   # standardised particles
    out["y"] = y.astype(np.float32)
    out["r"] = rs.normal(size=(B, inf_dim(case))).astype(np.float32)
    out["mask"] = circular_mask(n, m) if case["mask"] else None
    if case["ctf"]:
        k = n - 1 if n % 2 == 0 else n                               # train_particles.py:355-358
        f = rs.normal(size=(B, 1, k, k)) / k
        f[:, 0, k // 2, k // 2] += 1.0
        out["ctf"] = f.astype(np.float32)
    else:
        out["ctf"] = None
    # augmentation angles: what np.random.uniform(0, 2*pi, size=B) returns after np.random.seed(seed)
    out["offset"] = np.random.RandomState(case["seed"]).uniform(0, 2 * np.pi, size=B) if case["augment"] else None
    # decoder-only entry SpatialGenerator.forward(x, z): explicit coords, z, upstream grad
    out["dec_x"] = (rs.uniform(-1.3, 1.3, size=(B, N, 2))).astype(np.float32)
    out["dec_z"] = rs.normal(size=(B, case["z_dim"])).astype(np.float32)
    out["dec_dy"] = rs.normal(size=(B, N, case["n_out"])).astype(np.float32)
    return out


def sample_idx(size, count=4096):
    """Strided sample used by store='sampled' cases."""
    if size <= count:
        return np.arange(size)
    return np.linspace(0, size - 1, count).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------
# Training-LOOP fixtures (tests/golden/gen_epoch_golden.py runs the reference's own train_epoch:
# train_mnist.py:127-171, train_particles.py:151-202) and --vanilla fixtures (VanillaGenerator through
# eval_minibatch: models.py:135-172, train_mnist.py:351-357, train_particles.py:446-452).
# An epoch case = a decoder/encoder case + the minibatch sizes of each epoch (a ragged last batch on purpose), the
# z_scale of each epoch (--z-delay: 0 for the first epochs, train_particles.py:500-504) and the Adam learning rate.
def _epoch(name, base, batches, z_scales=(1,), lr=1e-3):
    case = dict(base, name=name)
    return dict(name=name, case=case, batches=tuple(batches), z_scales=tuple(z_scales), lr=lr)


EPOCH_CASES = [
    _epoch("epoch_mnist", _case("_", n=7, m=7, H=20, L=2, seed=60), batches=(4, 4, 3)),
    _epoch("epoch_particles_zdelay", _case("_", script="particles", n=6, m=6, z_dim=3, H=24, L=2, theta_prior=np.pi, seed=61),
           batches=(5, 5, 2), z_scales=(0, 1)),
    _epoch("epoch_particles_ctf_mask", _case("_", script="particles", n=8, m=8, z_dim=2, H=24, L=3, ctf=True, mask=True,
                                               theta_prior=np.pi, seed=62), batches=(6, 3)),
]
EPOCH_CASES_BY_NAME = {e["name"]: e for e in EPOCH_CASES}


def build_epoch_inputs(ec):
    """Initial parameters, the per-epoch minibatches (and CTF filters) and the noise of every step."""
    case = ec["case"]
    rs = np.random.RandomState(2000 + case["seed"])
    n, m = case["n"], case["m"]
    N = n * m
    out = dict(x_coord=coord_grid(n, m), p_state=generator_state(case, rs), q_state=inference_state(case, rs),
               mask=circular_mask(n, m) if case["mask"] else None)
    total = sum(ec["batches"])
    if case["script"] == "mnist":
        u = rs.uniform(size=(total, N))
        y = np.floor(u * (rs.uniform(size=(total, N)) > 0.6) * 255.0) / 255.0
    else:
        y = rs.normal(size=(total, N))
    out["y"] = y.astype(np.float32)
    if case["ctf"]:
        k = n - 1 if n % 2 == 0 else n
        f = rs.normal(size=(total, 1, k, k)) / k
        f[:, 0, k // 2, k // 2] += 1.0
        out["ctf"] = f.astype(np.float32)
    else:
        out["ctf"] = None
    # one noise tensor per step: epochs x minibatches
    out["r"] = [[rs.normal(size=(b, inf_dim(case))).astype(np.float32) for b in ec["batches"]] for _ in ec["z_scales"]]
    # the noise of the evaluation pass that follows the last training epoch (eval_model draws too: train_mnist.py:174-226)
    out["r_eval"] = [rs.normal(size=(b, inf_dim(case))).astype(np.float32) for b in ec["batches"]]
    return out


def _vanilla(name, script="mnist", n=6, m=6, B=5, z_dim=3, H=20, L=2, act="tanh", n_out=1, softplus=False, resid=False,
             q_hidden=24, q_layers=1, theta_prior=np.pi, mask=False, seed=0):
    return dict(name=name, script=script, n=n, m=m, B=B, z_dim=z_dim, H=H, L=L, act=act, n_out=n_out, softplus=softplus,
                resid=resid, q_hidden=q_hidden, q_layers=q_layers, theta_prior=float(theta_prior), mask=mask, seed=seed,
                rotate=False, translate=False, dx_scale=0.1, z_scale=1.0, expand_coords=False, bilinear=False, ctf=False,
                wscale=1.0, store="full", augment=False)


VANILLA_CASES = [
    _vanilla("vanilla_mnist", seed=70),                                                        # train_mnist.py --vanilla
    _vanilla("vanilla_mnist_leaky_L3", L=3, act="leakyrelu", seed=71),
    _vanilla("vanilla_particles", script="particles", seed=72),                                # Gaussian, unit variance
    _vanilla("vanilla_particles_softplus_noise", script="particles", n_out=2, softplus=True, resid=True, L=3, mask=True,
             seed=73),                                                                         # --fit-noise --softplus --resid
]
VANILLA_CASES_BY_NAME = {c["name"]: c for c in VANILLA_CASES}


def vanilla_state(case, rs):
    """Parameters of VanillaGenerator in construction order (models.py:146-157)."""
    H, L, zd = case["H"], case["L"], case["z_dim"]
    st = OrderedDict()
    w, b = _linear(rs, H, zd, 1.0)
    st["layers.0.weight"], st["layers.0.bias"] = w, b
    idx = 2
    for _ in range(1, L):
        w, b = _linear(rs, H, H, 1.0)
        if case["resid"]:
            st["layers.%d.linear.weight" % idx], st["layers.%d.linear.bias" % idx] = w, b
            idx += 1
        else:
            st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = w, b
            idx += 2
    w, b = _linear(rs, case["n"] * case["m"] * case["n_out"], H, 1.0)
    st["layers.%d.weight" % idx], st["layers.%d.bias" % idx] = w, b
    return st


def build_vanilla_inputs(case):
    rs = np.random.RandomState(3000 + case["seed"])
    n, m, B = case["n"], case["m"], case["B"]
    N = n * m
    out = dict(x_coord=coord_grid(n, m), p_state=vanilla_state(case, rs), q_state=inference_state(case, rs),
               mask=circular_mask(n, m) if case["mask"] else None)
    if case["script"] == "mnist":
        y = np.floor(rs.uniform(size=(B, N)) * (rs.uniform(size=(B, N)) > 0.6) * 255.0) / 255.0
    else:
        y = rs.normal(size=(B, N))
    out["y"] = y.astype(np.float32)
    out["r"] = rs.normal(size=(B, case["z_dim"])).astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Seeded end-to-end runs (gen_seeded_golden.py): the reference's main() flow -- default initialisation, DataLoader(shuffle=True),
# un-patched noise draws -- under torch.manual_seed(seed) / np.random.seed(seed), against the build's command lines with
# `--seed seed --synthetic count`.  `argv` is the build's command line (this fork's flag spellings: underscores for
# mnist/galaxy, hyphens for particles); the generator reads the same numbers from the fields beside it.
def synthetic_images(kind, count, n, m, channels, seed):
    """The `--synthetic` dataset of the three command lines (spatial_vae_amd/cli.py synthetic_images; a CPU test keeps the two
    definitions equal)."""
    rs = np.random.RandomState(seed)
    if kind == "particles":
        return rs.normal(size=(count, n, m)).astype(np.float32)
    shape = (count, n, m) if channels == 1 else (count, n, m, channels)
    u = rs.uniform(size=shape)
    keep = rs.uniform(size=shape) > (0.8 if channels == 1 else 0.0)
    return np.floor(u * keep * 255.0).astype(np.float32)


SEEDED_CASES = [
    dict(name="seeded_mnist", script="mnist", seed=1234, count=300, n=28, m=28, channels=1, z_dim=2, H=64, q_hidden=32, L=2,
         q_layers=2, bs=64, epochs=2, lr=1e-3, save_interval=1, theta_prior=np.pi / 4, dx_scale=0.1, z_delay=0, augment=False,
         argv=["--synthetic", "300", "--num_epochs", "2", "--minibatch_size", "64", "--p_hidden_dim", "64", "--q_hidden_dim", "32",
               "-l", "1e-3", "--save_prefix", "s", "--save_interval", "1", "--progress_every", "0", "--seed", "1234"]),
    dict(name="seeded_galaxy", script="galaxy", seed=77, count=48, n=32, m=32, channels=3, z_dim=4, H=32, q_hidden=32, L=3,
         q_layers=1, bs=20, epochs=2, lr=1e-3, save_interval=1, theta_prior=np.pi, dx_scale=0.1, z_delay=1, augment=True,
         argv=["x", "y", "--synthetic", "48", "--num_epochs", "2", "--minibatch_size", "20", "-z", "4", "--p_hidden_dim", "32",
               "--q_hidden_dim", "32", "--p_num_layers", "3", "--q_num_layers", "1", "-l", "1e-3", "--z_delay", "1",
               "--augment_rotation",
               "--save_prefix", "s", "--save_interval", "1", "--progress_every", "0", "--seed", "77"]),
    dict(name="seeded_particles", script="particles", seed=4321, count=96, n=40, m=40, channels=1, z_dim=3, H=48, q_hidden=32,
         L=2, q_layers=1, bs=40, epochs=2, lr=1e-3, save_interval=10, theta_prior=np.pi, dx_scale=0.1, z_delay=1, augment=False,
         fit_noise=True,
         argv=["x", "y", "--synthetic", "96", "--num-epochs", "2", "--minibatch-size", "40", "-z", "3", "--p-hidden-dim", "48",
               "--q-hidden-dim", "32", "--p-num-layers", "2", "--q-num-layers", "1", "-l", "1e-3", "--z-delay", "1", "--fit-noise",
               "--progress-every",
               "0", "--seed", "4321"]),
]
SEEDED_CASES_BY_NAME = {c["name"]: c for c in SEEDED_CASES}
