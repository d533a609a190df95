#!/usr/bin/env python3
"""Train spatial-VAE on particle images -- MI355X build.  Same flags (hyphenated) as the reference's
train_particles.py (/root/reference/train_particles.py:275-320) and the same stdout table
(Epoch, Split, ELBO, Error, KL).  The loop lives in spatial_vae_amd/cli.py."""
import argparse
import sys

import numpy as np
import torch

import spatial_vae.models as models
from spatial_vae_amd import cli, ctf as C, mrc, ops


def particle_arguments(argv=None):
    p = argparse.ArgumentParser("Train spatial-VAE on particle datasets")
    p.add_argument("train_path")
    p.add_argument("test_path")
    p.add_argument("--ctf-train")
    p.add_argument("--ctf-test")
    p.add_argument("--scale", default=1, type=float)
    p.add_argument("-z", "--z-dim", type=int, default=2)
    p.add_argument("--p-hidden-dim", type=int, default=500)
    p.add_argument("--p-num-layers", type=int, default=2)
    p.add_argument("--q-hidden-dim", type=int, default=500)
    p.add_argument("--q-num-layers", type=int, default=2)
    p.add_argument("-a", "--activation", choices=["tanh", "relu"], default="tanh")
    p.add_argument("--softplus", action="store_true")
    p.add_argument("--resid", action="store_true")
    p.add_argument("--expand-coords", action="store_true")
    p.add_argument("--bilinear", action="store_true")
    p.add_argument("--fit-noise", action="store_true")
    p.add_argument("--vanilla", action="store_true")
    p.add_argument("--no-rotate", action="store_true")
    p.add_argument("--no-translate", action="store_true")
    p.add_argument("--dx-scale", type=float, default=0.1)
    p.add_argument("--theta-prior", type=float, default=np.pi)
    p.add_argument("-l", "--learning-rate", type=float, default=1e-4)
    p.add_argument("--minibatch-size", type=int, default=100)
    p.add_argument("--augment-rotation", action="store_true")
    p.add_argument("--z-delay", type=int, default=0)
    p.add_argument("--normalize", action="store_true")
    p.add_argument("-c", "--crop", type=int, default=-1)
    p.add_argument("--save-prefix")
    p.add_argument("--save-interval", default=10, type=int)
    p.add_argument("--num-epochs", type=int, default=100)
    p.add_argument("-d", "--device", type=int, default=-2)
    p.add_argument("--no-preload", action="store_true", help="do not preload data into GPU RAM: the dataset stays in host "
                   "memory and each minibatch is uploaded")
    p.add_argument("--mask", action="store_true")
    p.add_argument("--synthetic", type=int, default=0, help="train on this many synthetic 40x40 particles (paths ignored)")
    p.add_argument("--progress-every", type=int, default=50)
    p.add_argument("--seed", type=int, default=None,
                   help="seed torch and numpy before the networks are built (the reference has no such flag: unseeded by default)")
    p.add_argument("--gemm", choices=["fp32", "fp16x3"], default=None,
                   help="hidden-layer GEMM path (default: SVAE_GEMM or fp32 MFMA; fp16x3 = fp32-accurate split-operand f16 MFMA)")
    return p.parse_args(argv)


def load_images(path):
    """train_particles.py:248-256: an MRC/MRCS stack (memory-mapped here) or a .npy array."""
    if path.endswith("mrc") or path.endswith("mrcs"):
        return mrc.read(path)[0]
    if path.endswith("npy"):
        return np.load(path, mmap_mode="r")
    raise SystemExit("particle stacks are read from .mrc/.mrcs or .npy files: " + path)


def build(args, device):
    if args.synthetic > 0:
        tr = cli.synthetic_images("particles", args.synthetic, 40, 40, 1, 0)
        te = cli.synthetic_images("particles", max(args.synthetic // 4, 1), 40, 40, 1, 1)
    else:
        tr, te = load_images(args.train_path), load_images(args.test_path)
    if args.crop > 0:                                               # centre crop (spatial_vae/image.py crop)
        def crop(a, c):
            n, m = a.shape[1:]
            i, j = (n - c) // 2, (m - c) // 2
            return a[:, i:i + c, j:j + c]
        tr, te = crop(tr, args.crop), crop(te, args.crop)
    n, m = tr.shape[1:]
    if args.normalize:                                              # train_particles.py:339-347
        def norm(a):
            flat = a.reshape(-1, n * m)
            return (a - flat.mean(1)[:, None, None]) / flat.std(1)[:, None, None]
        tr, te = norm(tr), norm(te)
    kn, km = (n - 1 if n % 2 == 0 else n), (m - 1 if m % 2 == 0 else m)   # train_particles.py:352-358
    ctf_train = ctf_test = None
    if args.ctf_train is not None:
        ctf_train = ops.ctf_filter(C.ctf_table(C.parse_ctf(args.ctf_train)), kn, km, scale=args.scale, device=device).unsqueeze(1)
    if args.ctf_test is not None:
        ctf_test = ops.ctf_filter(C.ctf_table(C.parse_ctf(args.ctf_test)), kn, km, scale=args.scale, device=device).unsqueeze(1)
    y_train = torch.from_numpy(np.ascontiguousarray(tr)).float().view(-1, n * m)
    y_test = torch.from_numpy(np.ascontiguousarray(te)).float().view(-1, n * m)
    mask = None
    if args.mask:                                                   # train_particles.py:384-392
        radius = min(n, m) / 2
        yg, xg = np.ogrid[:n, :m]
        dist = np.sqrt((n / 2 - yg) ** 2 + (m / 2 - xg) ** 2)
        mask = (torch.from_numpy(dist) < radius).view(-1)
        print("# masking to size:", int(mask.sum()), file=sys.stderr)
    act = cli.activation_class("particles", args.activation)
    n_out = 2 if args.fit_noise else 1
    if args.vanilla:
        p_net = models.VanillaGenerator(n * m, args.z_dim, args.p_hidden_dim, n_out=n_out, num_layers=args.p_num_layers,
                                        activation=act, softplus=args.softplus, resid=args.resid)
        rotate = translate = False
        inf_dim = args.z_dim
    else:
        rotate, translate = not args.no_rotate, not args.no_translate
        inf_dim = args.z_dim + (1 if rotate else 0) + (2 if translate else 0)
        p_net = models.SpatialGenerator(args.z_dim, args.p_hidden_dim, n_out=n_out, num_layers=args.p_num_layers, activation=act,
                                        softplus=args.softplus, resid=args.resid, expand_coords=args.expand_coords,
                                        bilinear=args.bilinear)
    q_net = models.InferenceNetwork(n * m, inf_dim, args.q_hidden_dim, num_layers=args.q_num_layers, activation=act,
                                    resid=args.resid)
    return dict(y_train=y_train, y_test=y_test, ctf_train=ctf_train, ctf_test=ctf_test, mask=mask, n=n, m=m, p_net=p_net,
                q_net=q_net, rotate=rotate, translate=translate, augment=args.augment_rotation,
                table=["Epoch", "Split", "ELBO", "Error", "KL"])


if __name__ == "__main__":
    sys.exit(cli.train_main("particles", particle_arguments(), build))
