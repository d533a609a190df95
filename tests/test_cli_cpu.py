"""CPU checks of the command-line layer: flag surfaces equal to the reference's,
the running-mean metric arithmetic, the host CTF filter generator against a fixture made by the reference."""
import os

import numpy as np
import torch

from helpers import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dests(ns):
    return set(vars(ns).keys())


def test_mnist_flag_surface_and_defaults():
    import train_mnist
    a = train_mnist.mnist_arguments(["--save_prefix", "p"])
    # /root/reference/train_mnist.py:232-263 (underscore spellings)
    want = {"dataset", "z_dim", "p_hidden_dim", "q_hidden_dim", "num_layers", "activation", "vanilla", "no_rotate",
            "no_translate", "dx_scale", "theta_prior", "learning_rate", "minibatch_size", "save_prefix", "save_interval",
            "num_epochs", "device", "num_train_images", "val_split"}
    assert want <= _dests(a)
    assert (a.dataset, a.z_dim, a.p_hidden_dim, a.q_hidden_dim, a.num_layers, a.activation) == \
        ("mnist-rotated-translated", 2, 500, 500, 2, "tanh")
    assert abs(a.theta_prior - np.pi / 4) < 1e-12 and a.dx_scale == 0.1 and a.learning_rate == 1e-4
    assert (a.minibatch_size, a.save_interval, a.num_epochs, a.device) == (100, 10, 100, -2)


def test_galaxy_flag_surface_and_defaults():
    import train_galaxy
    a = train_galaxy.galaxy_arguments(["tr.npy", "te.npy"])
    want = {"train_path", "test_path", "z_dim", "p_hidden_dim", "p_num_layers", "q_hidden_dim", "q_num_layers", "activation",
            "vanilla", "no_rotate", "no_translate", "dx_scale", "theta_prior", "learning_rate", "minibatch_size",
            "augment_rotation", "z_delay", "save_prefix", "save_interval", "num_epochs", "device", "num_train_images",
            "val_split", "make_mono", "logging_level", "invert_colours"}
    assert want <= _dests(a)
    assert (a.q_hidden_dim, a.p_hidden_dim, a.p_num_layers, a.q_num_layers) == (5000, 500, 2, 2)   # train_galaxy.py:304-307
    assert abs(a.theta_prior - np.pi) < 1e-12


def test_particles_flag_surface_uses_hyphens():
    import train_particles
    a = train_particles.particle_arguments(["tr.npy", "te.npy", "--fit-noise", "--no-translate", "--z-dim", "8", "--ctf-train", "t.txt",
                                            "--p-hidden-dim", "64", "--minibatch-size", "7", "--expand-coords", "--bilinear",
                                            "--resid", "--softplus", "--mask", "--normalize", "-c", "32"])
    assert a.fit_noise and a.no_translate and a.z_dim == 8 and a.ctf_train == "t.txt" and a.p_hidden_dim == 64
    assert a.minibatch_size == 7 and a.expand_coords and a.bilinear and a.resid and a.softplus and a.mask and a.crop == 32
    assert abs(a.theta_prior - np.pi) < 1e-12 and a.scale == 1


def test_activation_maps_follow_the_scripts_quirks():
    import torch.nn as nn
    from spatial_vae_amd import cli
    assert cli.activation_class("mnist", "relu") is nn.LeakyReLU          # train_mnist.py:344-348
    assert cli.activation_class("particles", "relu") is nn.LeakyReLU      # train_particles.py:433-436
    assert cli.activation_class("galaxy", "relu") is nn.ReLU              # train_galaxy.py:431-432
    assert cli.activation_class("galaxy", "leakyrelu") is nn.Tanh         # the reference's mis-spelt branch, :429


def test_running_mean_matches_reference_arithmetic():
    from spatial_vae_amd.cli import RunningMean
    rs = np.random.RandomState(0)
    sizes = [100, 100, 37]
    vals = rs.normal(size=(3, 3))
    acc, count = np.zeros(3), 0
    rm = RunningMean(torch.device("cpu"))
    for b, v in zip(sizes, vals):
        count += b
        ref = v * np.array([1.0, -1.0, 1.0])                                # the reference logs gen_loss = -log_p
        acc += b * (ref - acc) / count                                      # train_mnist.py:156-164
        rm.update(b, torch.tensor(v))                                       # (elbo, log_p, kl) as eval_minibatch returns them
    assert rm.seen == sum(sizes) and rm.count == 0                          # nothing is computed (or synchronised) per step
    assert np.array_equal(np.array(rm.values()), acc)                       # bit for bit: negation commutes with the arithmetic
    assert rm.count == sum(sizes) and not rm.pending
    tail = torch.tensor([1.0, 2.0, 3.0])                                    # a buffer the next step overwrites (DP metric tail)
    rm2 = RunningMean()
    rm2.update(4, tail, volatile=True)
    tail.zero_()
    assert rm2.values() == [1.0, -2.0, 3.0]


def test_coord_grid_matches_case_builder():
    import cases as C
    from spatial_vae_amd.cli import coord_grid
    assert np.array_equal(coord_grid(9, 5).numpy(), C.coord_grid(9, 5))


def test_ctf_oracle_matches_reference_fixture():
    """oracle/ctf_oracle.py (what the device kernel is tested against) reproduces the filters the reference's own
    spatial_vae/ctf.py wrote, including sizes above the kernel's in-LDS limit."""
    from oracle import ctf_oracle as O
    from spatial_vae_amd import ctf as C
    with np.load(os.path.join(GOLDEN_DIR, "ctf_golden.npz")) as f:
        gold = {k: f[k] for k in f.files}
    params = C.parse_ctf(os.path.join(GOLDEN_DIR, "ctf_table.txt"))
    assert np.array_equal(C.ctf_table(params), gold["table"])
    for key, (n, m, s) in {"filt_9x9_s1": (9, 9, 1), "filt_15x13_s2": (15, 13, 2), "filt_39x39_s1": (39, 39, 1)}.items():
        got = O.ctf_filter(params, n, m, scale=s)
        assert got.shape == gold[key].shape
        assert np.abs(got - gold[key]).max() <= 1e-7 * np.abs(gold[key]).max()
    first2 = {k: v[:2] for k, v in params.items()}
    for key, (n, m, s) in {"big_129x129_s1": (129, 129, 1), "big_100x96_s2": (100, 96, 2)}.items():
        got = O.ctf_filter(first2, n, m, scale=s)
        assert np.abs(got - gold[key]).max() <= 1e-7 * np.abs(gold[key]).max()


def test_image_grid_layout_and_label():
    """make_grid/save_image restatement (src/misc_tools.py:30-39 -> torchvision.utils.save_image(nrow=rows, padding=3,
    pad_value=0.5)) and MiscTools.save_label (src/misc_tools.py:15-28)."""
    import argparse
    from spatial_vae_amd import cli
    imgs = np.zeros((5, 1, 4, 6), np.float32)
    for k in range(5):
        imgs[k] = (k + 1) / 10.0
    g = cli.image_grid(imgs, nrow=2)
    assert g.shape == (3 * 7 + 3, 2 * 9 + 3, 3) and g.dtype == np.uint8          # 3 rows x 2 columns of (4+3) x (6+3) cells
    assert (g[:3] == 128).all() and (g[:, :3] == 128).all()                      # 0.5*255+0.5 -> 128 padding
    assert (g[3:7, 3:9] == int(0.1 * 255 + 0.5)).all()                           # image 0
    assert (g[3:7, 12:18] == int(np.float32(0.2) * np.float32(255) + np.float32(0.5))).all()   # image 1, same row
    assert (g[10:14, 3:9] == int(np.float32(0.3) * np.float32(255) + np.float32(0.5))).all()   # image 2, next row
    assert (g[17:21, 12:18] == 128).all()                                         # the sixth cell stays blank
    one = cli.image_grid(imgs[:1], nrow=1)
    assert one.shape == (4, 6, 3)                                                 # a single image is not padded
    ns = argparse.Namespace(z_dim=2, p_hidden_dim=5, p_num_layers=3, q_num_layers=1, save_prefix="run", num_epochs=7)
    assert cli.save_label(ns) == "run_z2pnl3qnl1ep7"


def test_bench_flop_accounting_matches_baseline_md():
    """bench.py's per-config FLOP figures are BASELINE.md section 4's (F_fwd in GF: 25.24 / 100.95 / 412.9 / 8818 / 206.0)."""
    import bench
    want = {1: 25.24e9, 2: 100.95e9, 3: 412.9e9, 4: 8.818e12, 5: 206.0e9}
    for cid, f in want.items():
        cfg = bench.CONFIGS[cid]
        fwd, step, gemm = bench.decoder_flops(cfg, cfg["B"])
        assert abs(fwd - f) / f < 2e-3, (cid, fwd)
        assert step == 3 * fwd and gemm == 2.0 * cfg["B"] * cfg["n"] ** 2 * cfg["H"] ** 2
    assert bench.inf_dim(bench.CONFIGS[1]) == 3 and bench.inf_dim(bench.CONFIGS[5]) == 11


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without torchrun spawns two ranks itself; with no GPU here both fail at device selection and
    the parent must come back non-zero instead of hanging (on the GPU box tests/test_gpu_bench.py sees the JSON line)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    if __import__("torch").cuda.is_available():
        return
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


# ---------------------------------------------------------------------------------------------------- seeded runs (row R)
def test_synthetic_dataset_definitions_agree():
    import cases as C
    from spatial_vae_amd import cli
    for kind, n, ch in (("mnist", 28, 1), ("galaxy", 32, 3), ("particles", 40, 1)):
        assert np.array_equal(cli.synthetic_images(kind, 7, n, n, ch, 3), C.synthetic_images(kind, 7, n, n, ch, 3))


def test_loader_order_is_the_dataloaders():
    """cli.loader_order consumes the global CPU generator exactly as iter(DataLoader) does and yields its order."""
    from spatial_vae_amd import cli
    data = torch.utils.data.TensorDataset(torch.arange(103))
    for shuffle in (True, False):
        torch.manual_seed(99)
        want = torch.cat([b for b, in torch.utils.data.DataLoader(data, batch_size=10, shuffle=shuffle)])
        after = torch.empty(5).normal_()
        torch.manual_seed(99)
        got = cli.loader_order(103, shuffle)
        assert torch.equal(got, want)
        assert torch.equal(torch.empty(5).normal_(), after)               # and leaves the generator in the same state


def _replay_seeded_on_cpu(case):
    """The build's random plan (cli.train_pass_plan / eval_pass_plan: shuffle + noise in the reference's order) driving the
    torch-CPU port of the reference step, from the seed alone."""
    import contextlib
    import io
    import cases as C
    import spatial_vae.models as models
    from oracle import torch_cpu_step as T
    from spatial_vae_amd import cli
    script, n, m = case["script"], case["n"], case["m"]
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    tr = cli.synthetic_images(script, case["count"], n, m, 1, 0)
    te = cli.synthetic_images(script, max(case["count"] // 4, 1), n, m, 1, 1)
    scale = 255.0 if script == "mnist" else 1.0
    y_train = torch.from_numpy(tr).float().div(scale).view(-1, n * m)
    y_test = torch.from_numpy(te).float().div(scale).view(-1, n * m)
    inf_dim = case["z_dim"] + 3
    with contextlib.redirect_stdout(io.StringIO()):
        p_net = models.SpatialGenerator(case["z_dim"], case["H"], n_out=2 if case.get("fit_noise") else 1, num_layers=case["L"])
        q_net = models.InferenceNetwork(n * m, inf_dim, case["q_hidden"], num_layers=case["q_layers"])
    init_sum = float(sum(p.detach().double().sum() for p in list(p_net.parameters()) + list(q_net.parameters())))
    trainer = T.CpuTrainer({k: v.detach().numpy() for k, v in p_net.state_dict().items()},
                           {k: v.detach().numpy() for k, v in q_net.state_dict().items()}, cli.coord_grid(n, m), lr=case["lr"],
                           script=script, act="tanh", rotate=True, translate=True, dx_scale=case["dx_scale"],
                           theta_prior=case["theta_prior"])
    cpu = torch.device("cpu")
    if script != "particles":
        cli.loader_order(len(y_test), False)                              # sample_images
    steps, rows = [], []
    for epoch in range(case["epochs"]):
        kw = {} if script == "mnist" else {"z_scale": 0 if epoch < case["z_delay"] else 1}
        mean = cli.RunningMean()
        batches, noise = cli.train_pass_plan(len(y_train), case["bs"], inf_dim, cpu)
        for idx, r in zip(batches, noise):
            out = trainer.step(y_train[idx], r, **kw)
            steps.append([float(v) for v in out])
            mean.update(idx.numel(), torch.stack(out))
        row = mean.values()
        shapes = None
        if script != "particles" and (epoch + 1) % case["save_interval"] == 0:
            shapes = cli.display_draw_shapes(script, inf_dim, case["z_dim"])
        tb, noise, _ = cli.eval_pass_plan(len(y_test), case["bs"], inf_dim, cpu, None, shapes)
        mean = cli.RunningMean()
        for idx, r in zip(tb, noise):
            with torch.no_grad():
                out = trainer.fn(trainer.pp, trainer.qp, trainer.x, y_test[idx], r, **dict(trainer.cfg, **kw))[:3]
            steps.append([float(v) for v in out])
            mean.update(idx.numel(), torch.stack(out))
        rows.append(row + mean.values())
    return init_sum, np.array(steps), np.array(rows), trainer


def test_seeded_runs_follow_the_reference_from_the_seed_alone():
    """Row R of the coverage table: with torch.manual_seed(s) the build consumes randomness in the reference's order --
    default initialisation (p_net, q_net), the sample-image pass, DataLoader(shuffle=True)'s two draws per epoch, one N(0,1)
    draw per minibatch on the CPU generator, the validation loader's draw, the display helpers' draws on dump epochs -- so the
    torch-CPU port driven by cli's plan reproduces the reference's un-patched seeded run (tests/golden/gen_seeded_golden.py):
    every step's (elbo, log_p, kl), the printed rows and the final parameters."""
    import cases as C
    from helpers import load_golden, rel_err
    for name in ("seeded_mnist", "seeded_particles"):
        case = C.SEEDED_CASES_BY_NAME[name]
        gold = load_golden(name)
        init_sum, steps, rows, trainer = _replay_seeded_on_cpu(case)
        assert abs(init_sum - float(gold["init_sum"])) <= 1e-9 * max(1.0, abs(float(gold["init_sum"]))), name
        assert steps.shape == gold["steps"].shape, name
        assert np.abs(steps - gold["steps"]).max() <= 2e-5 * np.abs(gold["steps"]).max(), (name, steps, gold["steps"])
        assert np.abs(rows - gold["rows"]).max() <= 2e-5 * np.abs(gold["rows"]).max(), name
        for k, v in trainer.pp.items():
            assert rel_err(v.detach().numpy(), gold["p." + k]) < 1e-4, (name, k)
        for k, v in trainer.qp.items():
            assert rel_err(v.detach().numpy(), gold["q." + k]) < 1e-4, (name, k)
