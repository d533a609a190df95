#!/usr/bin/env python3
"""Train spatial-VAE on galaxy images (RGB) -- MI355X build.  Same flags as the reference's train_galaxy.py
(/root/reference/train_galaxy.py:297-343).  The loop lives in spatial_vae_amd/cli.py."""
import argparse
import sys

import numpy as np
import torch

import spatial_vae.models as models
from spatial_vae_amd import cli


def galaxy_arguments(argv=None):
    p = argparse.ArgumentParser("Train spatial-VAE on galaxy datasets")
    p.add_argument("train_path")
    p.add_argument("test_path")
    p.add_argument("-z", "--z_dim", type=int, default=2)
    p.add_argument("--p_hidden_dim", type=int, default=500)
    p.add_argument("--p_num_layers", type=int, default=2)
    p.add_argument("--q_hidden_dim", type=int, default=5000)
    p.add_argument("--q_num_layers", type=int, default=2)
    p.add_argument("-a", "--activation", choices=["tanh", "relu", "leakyrelu", "sigmoid"], default="tanh")
    p.add_argument("--vanilla", action="store_true")
    p.add_argument("--no_rotate", action="store_true")
    p.add_argument("--no_translate", action="store_true")
    p.add_argument("--dx_scale", type=float, default=0.1)
    p.add_argument("--theta_prior", type=float, default=np.pi)
    p.add_argument("-l", "--learning_rate", type=float, default=1e-4)
    p.add_argument("--minibatch_size", type=int, default=100)
    p.add_argument("--augment_rotation", action="store_true")
    p.add_argument("--z_delay", type=int, default=0)
    p.add_argument("--save_prefix")
    p.add_argument("--save_interval", default=10, type=int)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("-d", "--device", type=int, default=-2)
    p.add_argument("--num_train_images", type=int, default=0)
    p.add_argument("--val_split", type=int, default=50)
    p.add_argument("--make_mono", action="store_true")
    p.add_argument("--logging_level", type=str, default="INFO")
    p.add_argument("--invert_colours", action="store_true")
    p.add_argument("--synthetic", type=int, default=0, help="train on this many synthetic 32x32x3 images (paths ignored)")
    p.add_argument("--progress_every", type=int, default=50)
    p.add_argument("--seed", type=int, default=None,
                   help="seed torch and numpy before the networks are built (the reference has no such flag: unseeded by default)")
    p.add_argument("--gemm", choices=["fp32", "fp16x3"], default=None,
                   help="hidden-layer GEMM path (default: SVAE_GEMM or fp32 MFMA; fp16x3 = fp32-accurate split-operand f16 MFMA)")
    return p.parse_args(argv)


def build(args, device):
    if args.synthetic > 0:
        tr = cli.synthetic_images("galaxy", args.synthetic, 32, 32, 3, 0)
        va = cli.synthetic_images("galaxy", max(args.synthetic // 4, 1), 32, 32, 3, 1)
    else:
        tr, va = np.load(args.train_path), np.load(args.test_path)
    channels = 3
    if args.make_mono:                                              # train_galaxy.py:366-370 (training images only)
        tr = np.mean(tr, axis=3)
        channels = 1
    np.random.shuffle(tr)
    if args.num_train_images > 0:
        tr, va = tr[:args.num_train_images], va[:args.num_train_images]
    n, m = tr.shape[1:3]
    y_train = torch.from_numpy(np.ascontiguousarray(tr)).float().div(255)
    y_val = torch.from_numpy(np.ascontiguousarray(va)).float().div(255)
    if args.invert_colours:
        y_train, y_val = 1 - y_train, 1 - y_val
    y_train, y_val = y_train.view(-1, n * m, channels), y_val.view(-1, n * m, channels)
    act = cli.activation_class("galaxy", args.activation)
    print("# training with z-dim:", args.z_dim, file=sys.stderr)
    if args.vanilla:
        p_net = models.VanillaGenerator(channels * n * m, args.z_dim, args.p_hidden_dim, num_layers=args.p_num_layers, activation=act)
        rotate = translate = False
        inf_dim = args.z_dim
    else:
        rotate, translate = not args.no_rotate, not args.no_translate
        inf_dim = args.z_dim + (1 if rotate else 0) + (2 if translate else 0)
        p_net = models.SpatialGenerator(args.z_dim, args.p_hidden_dim, n_out=channels, num_layers=args.p_num_layers, activation=act)
    q_net = models.InferenceNetwork(n * m * channels, inf_dim, args.q_hidden_dim, num_layers=args.q_num_layers, activation=act)
    return dict(y_train=y_train, y_test=y_val, n=n, m=m, p_net=p_net, q_net=q_net, rotate=rotate, translate=translate,
                augment=args.augment_rotation, table=["Epoch", "ELBO", "BCE loss", "KL"])


if __name__ == "__main__":
    sys.exit(cli.train_main("galaxy", galaxy_arguments(), build))
