#!/usr/bin/env python3
"""Train spatial-VAE on the MNIST variants -- MI355X build.  Same flags as the reference's train_mnist.py
(/root/reference/train_mnist.py:229-265: underscore spellings, theta prior pi/4), same stdout table
(Epoch, ELBO, BCE loss, KL).  The loop lives in spatial_vae_amd/cli.py."""
import argparse
import sys

import numpy as np
import torch

import spatial_vae.models as models
from spatial_vae_amd import cli


def mnist_arguments(argv=None):
    p = argparse.ArgumentParser("Train spatial-VAE on MNIST datasets")
    p.add_argument("--dataset", choices=["mnist", "mnist-rotated", "mnist-rotated-translated", "galaxy"],
                   default="mnist-rotated-translated")
    p.add_argument("-z", "--z_dim", type=int, default=2)
    p.add_argument("--p_hidden_dim", type=int, default=500)
    p.add_argument("--q_hidden_dim", type=int, default=500)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("-a", "--activation", choices=["tanh", "relu"], default="tanh")
    p.add_argument("--vanilla", action="store_true")
    p.add_argument("--no_rotate", action="store_true")
    p.add_argument("--no_translate", action="store_true")
    p.add_argument("--dx_scale", type=float, default=0.1)
    p.add_argument("--theta_prior", type=float, default=np.pi / 4)
    p.add_argument("-l", "--learning_rate", type=float, default=1e-4)
    p.add_argument("--minibatch_size", type=int, default=100)
    p.add_argument("--save_prefix")
    p.add_argument("--save_interval", default=10, type=int)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("-d", "--device", type=int, default=-2)
    p.add_argument("--num_train_images", type=int, default=0)      # parsed and unused, as in the reference
    p.add_argument("--val_split", type=int, default=50)            # parsed and unused, as in the reference
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="train on this many synthetic 28x28 images (no data files)")
    p.add_argument("--progress_every", type=int, default=50, help="stderr progress line every N steps (0 = never)")
    p.add_argument("--seed", type=int, default=None,
                   help="seed torch and numpy before the networks are built (the reference has no such flag: unseeded by default)")
    p.add_argument("--gemm", choices=["fp32", "fp16x3"], default=None,
                   help="hidden-layer GEMM path (default: SVAE_GEMM or fp32 MFMA; fp16x3 = fp32-accurate split-operand f16 MFMA)")
    return p.parse_args(argv)


def dataset_files(dataset):
    """(train, test) .npy paths of --dataset as the reference spells them (train_mnist.py:285-299): the MNIST variants are
    data/<name>/images_{train,test}.npy, galaxy zoo is data/galaxy_zoo/galaxy_zoo_{train,test}.npy."""
    if dataset == "galaxy":
        return "data/galaxy_zoo/galaxy_zoo_train.npy", "data/galaxy_zoo/galaxy_zoo_test.npy"
    sub = {"mnist-rotated": "mnist_rotated", "mnist-rotated-translated": "mnist_rotated_translated"}[dataset]
    return "data/{}/images_train.npy".format(sub), "data/{}/images_test.npy".format(sub)


def build(args, device):
    if args.synthetic > 0:
        tr = cli.synthetic_images("mnist", args.synthetic, 28, 28, 1, 0)
        te = cli.synthetic_images("mnist", max(args.synthetic // 4, 1), 28, 28, 1, 1)
    elif args.dataset == "mnist":
        raise SystemExit("--dataset mnist downloads through torchvision, which is not available here; "
                         "use the .npy datasets or --synthetic")
    else:
        tr, te = (np.load(f) for f in dataset_files(args.dataset))
        if args.dataset == "galaxy":                                # mono-chromed galaxy zoo: channel mean (train_mnist.py:297-301)
            tr, te = np.mean(tr, axis=3), np.mean(te, axis=3)
    n, m = tr.shape[1:3]
    y_train = torch.from_numpy(tr).float().div(255).view(-1, n * m)
    y_test = torch.from_numpy(te).float().div(255).view(-1, n * m)
    act = cli.activation_class("mnist", args.activation)
    print("# training with z-dim:", args.z_dim, file=sys.stderr)
    if args.vanilla:
        p_net = models.VanillaGenerator(n * m, args.z_dim, args.p_hidden_dim, num_layers=args.num_layers, activation=act)
        rotate = translate = False
        inf_dim = args.z_dim
    else:
        rotate, translate = not args.no_rotate, not args.no_translate
        inf_dim = args.z_dim + (1 if rotate else 0) + (2 if translate else 0)
        p_net = models.SpatialGenerator(args.z_dim, args.p_hidden_dim, n_out=1, num_layers=args.num_layers, activation=act)
    q_net = models.InferenceNetwork(n * m, inf_dim, args.q_hidden_dim, num_layers=args.num_layers, activation=act)
    return dict(y_train=y_train, y_test=y_test, n=n, m=m, p_net=p_net, q_net=q_net, rotate=rotate, translate=translate,
                table=["Epoch", "ELBO", "BCE loss", "KL"])


if __name__ == "__main__":
    sys.exit(cli.train_main("mnist", mnist_arguments(), build))
