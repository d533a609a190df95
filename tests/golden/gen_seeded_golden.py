#!/usr/bin/env python3
"""Seeded end-to-end golden runs, produced by running the REFERENCE on CPU (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_seeded_golden.py

For each case of cases.SEEDED_CASES the flow of the reference's main() (train_mnist.py:268-459, train_galaxy.py:346-557,
train_particles.py:272-543) is replayed with the reference's own pieces and NOTHING patched: torch.manual_seed(seed) and
np.random.seed(seed); the dataset (synthetic, cases.synthetic_images -- what the build's `--synthetic` flag loads) prepared as
main() prepares it (galaxy: np.random.shuffle of the training images, train_galaxy.py:372); SpatialGenerator then
InferenceNetwork with their DEFAULT initialisation; torch.optim.Adam over p_net then q_net parameters; real
torch.utils.data.DataLoader objects (shuffle=True for training); MiscTools.sample_images over the validation loader
(mnist / galaxy); then per epoch the reference's train_epoch and eval_model (with image dumps every save_interval epochs -- the
dump's extra noise draws are part of the random stream; torchvision's save_image is a stub, no file is written).  Noise comes
from x.data.new(B, z).normal_() on the CPU generator (train_mnist.py:38), augmentation angles from np.random
(train_galaxy.py:43-46) and Pillow.

Stored (data only): the (elbo, log_p, kl) of every training and evaluation minibatch in order, the per-epoch rows the script
prints (train and validation running means), a checksum of the initial parameters, and every parameter after the last epoch.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (loads the reference modules read-only, stubs torchvision / skimage)
import cases as C  # noqa: E402

from src.misc_tools import MiscTools  # noqa: E402  (the reference's, /root/reference is first on sys.path)


def run(case):
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"] % (2 ** 32))
    script = case["script"]
    n, m, ch = case["n"], case["m"], case["channels"]
    tr = C.synthetic_images(script, case["count"], n, m, ch, 0)
    te = C.synthetic_images(script, max(case["count"] // 4, 1), n, m, ch, 1)
    mod = {"mnist": G.ref_mnist, "galaxy": G.ref_galaxy, "particles": G.ref_particles}[script]
    if script == "galaxy":
        np.random.shuffle(tr)                                             # train_galaxy.py:372
        y_train = (torch.from_numpy(tr).float() / 255).view(-1, n * m, ch)
        y_test = (torch.from_numpy(te).float() / 255).view(-1, n * m, ch)
    elif script == "mnist":
        y_train = (torch.from_numpy(tr).float() / 255).view(-1, n * m)    # train_mnist.py:310-313
        y_test = (torch.from_numpy(te).float() / 255).view(-1, n * m)
    else:
        y_train = torch.from_numpy(tr).float().view(-1, n * m)            # train_particles.py:360-361
        y_test = torch.from_numpy(te).float().view(-1, n * m)
    x0, x1 = np.meshgrid(np.linspace(-1, 1, m), np.linspace(1, -1, n))
    x_coord = torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1)).float()
    data_train = torch.utils.data.TensorDataset(y_train)
    data_test = torch.utils.data.TensorDataset(y_test)
    inf_dim = case["z_dim"] + 3
    n_out = ch if script == "galaxy" else (2 if case.get("fit_noise") else 1)
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull                              # the constructors print(self)
    try:
        p_net = G.ref_models.SpatialGenerator(case["z_dim"], case["H"], n_out=n_out, num_layers=case["L"], activation=torch.nn.Tanh)
        q_net = G.ref_models.InferenceNetwork(n * m * (ch if script == "galaxy" else 1), inf_dim, case["q_hidden"],
                                              num_layers=case["q_layers"], activation=torch.nn.Tanh)
    finally:
        sys.stdout = stdout
    init_sum = float(sum(p.detach().double().sum() for p in list(p_net.parameters()) + list(q_net.parameters())))
    params = list(p_net.parameters()) + list(q_net.parameters())
    optim = torch.optim.Adam(params, lr=case["lr"])
    train_iterator = torch.utils.data.DataLoader(data_train, batch_size=case["bs"], shuffle=True)
    val_iterator = torch.utils.data.DataLoader(data_test, batch_size=case["bs"])
    if script != "particles":
        MiscTools.sample_images(iterator=val_iterator, image_dims=[n, m], output_dir="/nonexistent", save_label="l")

    steps = []
    orig_eval = mod.eval_minibatch

    def recording_eval(*a, **k):
        res = orig_eval(*a, **k)
        steps.append([float(res[0]), float(res[1]), float(res[2])])
        return res

    mod.eval_minibatch = recording_eval
    rows = []
    stderr, sys.stderr = sys.stderr, devnull                              # the progress line
    kw = dict(rotate=True, translate=True, dx_scale=case["dx_scale"], theta_prior=case["theta_prior"], use_cuda=False)
    try:
        for epoch in range(case["epochs"]):
            z_scale = 0 if epoch < case["z_delay"] else 1
            dump = (epoch + 1) % case["save_interval"] == 0
            if script == "mnist":
                tr_row = mod.train_epoch(train_iterator, x_coord, p_net, q_net, optim, epoch=epoch, num_epochs=case["epochs"],
                                         N=len(y_train), **kw)
                ev_row = mod.eval_model(val_iterator, x_coord, p_net, q_net, to_save_image_samples=dump, image_dims=[n, m],
                                        epoch=str(epoch + 1), output_dir="/nonexistent", save_label="l", **kw)
            elif script == "galaxy":
                tr_row = mod.train_epoch(train_iterator, x_coord, p_net, q_net, optim, augment_rotation=case["augment"],
                                         z_scale=z_scale, epoch=epoch, num_epochs=case["epochs"], train_images_len=len(y_train), **kw)
                ev_row = mod.eval_model(val_iterator, x_coord, p_net, q_net, z_dim=case["z_dim"], z_scale=z_scale,
                                        to_save_image_samples=dump, image_dims=[n, m], epoch=str(epoch + 1),
                                        output_dir="/nonexistent", save_label="l", **kw)
            else:
                tr_row = mod.train_epoch(train_iterator, x_coord, None, p_net, q_net, optim, augment_rotation=case["augment"],
                                         z_scale=z_scale, epoch=epoch, num_epochs=case["epochs"], N=len(y_train), **kw)
                ev_row = mod.eval_model(val_iterator, x_coord, None, p_net, q_net, z_scale=z_scale, **kw)
            rows.append([float(v) for v in tr_row] + [float(v) for v in ev_row])
    finally:
        sys.stderr = stderr
        mod.eval_minibatch = orig_eval
    out = {"steps": np.array(steps, np.float64), "rows": np.array(rows, np.float64), "init_sum": np.float64(init_sum)}
    for k, p in p_net.named_parameters():
        out["p." + k] = p.detach().numpy()
    for k, p in q_net.named_parameters():
        out["q." + k] = p.detach().numpy()
    return out


def main():
    torch.set_num_threads(4)
    for case in C.SEEDED_CASES:
        out = run(case)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **out)
        print("%-18s steps %d rows %s" % (case["name"], len(out["steps"]), np.round(out["rows"], 3).tolist()))


if __name__ == "__main__":
    main()
