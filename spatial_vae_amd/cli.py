"""Shared machinery of the three training command lines (train_mnist.py, train_galaxy.py,
train_particles.py at the repository root).

The reference triplicates its loop per script (train_mnist.py:127-226 and 268-469, train_galaxy.py:186-294
and 346-575, train_particles.py:151-245 and 272-547); here ONE loop serves all three and the scripts only
declare their flag surfaces (which differ: underscores for mnist/galaxy, hyphens for particles).  Kept:
flag names and defaults, the step order (loss = -elbo; backward; step; zero_grad), the running-mean metric
arithmetic, the stdout tables, train.txt / val.txt / command.txt / models.txt, and the whole-module
`.sav` checkpoints with the reference's file names.  Dropped (out of scope): the
interactive "clear outputs?" prompt (the run directory is created, never wiped), the loss-curve SVG, the zip
archive and dataset download.  `--augment_rotation` runs on the device
(ops.rotate_augment, bit-identical to the reference's per-image Pillow loop).  Added: `--synthetic N`
(train on N synthetic images when no data files exist), data-parallel execution under torchrun, and
`--progress_every` (the reference pays three .item() syncs per step for its progress line).
"""
import copy
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

from . import dp
from . import elbo as E


class RunningMean(object):
    """acc += b * (v - acc) / count  (train_mnist.py:156-164) in Python doubles, exactly as the reference does it -- but
    LAZILY: update() only keeps the minibatch's (elbo, log_p, kl) device vector, and values() fetches everything collected so far
    in one transfer and replays the arithmetic on the host.  No kernel and no synchronisation per step (the reference pays
    three .item() calls per step; a device-side accumulator paid five tiny kernels).  The reference logs -log_p ("gen_loss");
    negation commutes with this arithmetic bit for bit, so the sign is applied when the values are read."""

    def __init__(self, device=None, n=3):
        self.acc = [0.0] * n
        self.count = 0          # images whose metrics have been folded into acc
        self.seen = 0           # images handed to update() so far
        self.pending = []

    def update(self, batch_size, vec, volatile=False):
        """vec: the minibatch's 3-vector on the device.  volatile=True: the tensor will be overwritten by the next step (the
        data-parallel metric tail), so a copy is kept."""
        self.seen += batch_size
        self.pending.append((batch_size, vec.detach().clone() if volatile else vec.detach()))

    def _flush(self):
        if not self.pending:
            return
        host = torch.stack([v.reshape(-1) for _, v in self.pending]).cpu().double().tolist()   # the one synchronisation
        for (b, _), row in zip(self.pending, host):
            self.count += b
            self.acc = [a + b * (v - a) / self.count for a, v in zip(self.acc, row)]
        self.pending = []

    def values(self):
        self._flush()
        e, lp, k = self.acc
        return [e, -lp, k]


def coord_grid(n_rows, n_cols):
    """(N, 2) grid of train_mnist.py:315-320."""
    x0, x1 = np.meshgrid(np.linspace(-1, 1, n_cols), np.linspace(1, -1, n_rows))
    return torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1)).float()


def loader_order(n, shuffle):
    """Index order of one pass over torch.utils.data.DataLoader(dataset, batch_size, shuffle=shuffle) as the reference makes it
    (train_mnist.py:395-396, :138, :196), consuming torch's global CPU generator the way iter(DataLoader) does: the iterator
    draws one int64 (its worker base seed) whether or not the loader shuffles; RandomSampler then draws a second int64, seeds a
    private generator with it and takes torch.randperm(n) from that one."""
    torch.empty((), dtype=torch.int64).random_()
    if not shuffle:
        return torch.arange(n)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


def pass_noise(sizes, inf_dim, device, after_first=()):
    """The N(0,1) draws of one pass over minibatches of `sizes` images, from torch's global CPU generator in the reference's
    order and shapes: one x.data.new(B, inf_dim).normal_() per minibatch (train_mnist.py:38; CPU tensors in the CPU reference),
    and -- on a pass that dumps images -- the display helpers' draws right after the first minibatch's (`after_first`: their
    shapes; train_mnist.py:107, train_galaxy.py:146, :177).  The shapes matter: the CPU normal_ kernel fills in blocks of 16 and
    re-draws the tail, so one big draw is not the concatenation of the small ones.  Returns (per-minibatch device tensors,
    display draws on the device); everything is uploaded in ONE transfer, nothing is drawn or copied inside the step loop."""
    draws, extra = [], []
    for i, b in enumerate(sizes):
        draws.append(torch.empty(b, inf_dim).normal_())
        if i == 0:
            extra = [torch.empty(*shape).normal_() for shape in after_first]
    flat = torch.cat([t.reshape(-1) for t in draws + extra]) if draws else torch.empty(0)
    flat = flat.to(device, non_blocking=True)
    out, off = [], 0
    for t in draws + extra:
        out.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return out[:len(draws)], out[len(draws):]


def train_pass_plan(N, bs, inf_dim, device, home=None):
    """(index minibatches, their noise) of one training pass: iter(DataLoader(shuffle=True)) then one draw per minibatch
    (train_mnist.py:138-143 via :38).  The last minibatch is ragged, as the reference's loader keeps it (no drop_last)."""
    perm = loader_order(N, True)
    batches = [perm[i:i + bs].to(home if home is not None else device) for i in range(0, N, bs)]
    noise, _ = pass_noise([b.numel() for b in batches], inf_dim, device)
    return batches, noise


def eval_pass_plan(ntest, bs, inf_dim, device, home=None, display_shapes=()):
    """(index minibatches, their noise, the display helpers' noise) of one evaluation pass over the un-shuffled validation
    loader (train_mnist.py:196-201); on a dump epoch eval_model decodes the first minibatch again for the PNG files
    (train_mnist.py:214-224, train_galaxy.py:275-292) -- `display_shapes` maps a minibatch size to those draws' shapes."""
    order = loader_order(ntest, False)
    tb = [order[i:i + bs].to(home if home is not None else device) for i in range(0, ntest, bs)]
    shapes = display_shapes(tb[0].numel()) if (display_shapes and tb) else ()
    noise, shown = pass_noise([b.numel() for b in tb], inf_dim, device, after_first=shapes)
    return tb, noise, shown


def activation_class(script, name):
    """The scripts' (inconsistent) flag-to-module maps: 'relu' means LeakyReLU in mnist/particles
    (train_mnist.py:344-348, train_particles.py:433-436) but nn.ReLU in galaxy, where 'leakyrelu' is
    mis-spelt in the reference and silently gives Tanh (train_galaxy.py:426-434)."""
    if script == "galaxy":
        return {"tanh": nn.Tanh, "relu": nn.ReLU, "sigmoid": nn.Sigmoid}.get(name, nn.Tanh)
    return nn.Tanh if name == "tanh" else nn.LeakyReLU


def pick_device(d, world=1, local=0):
    """-d/--device: -1 = CPU, >= 0 = that GPU, -2 (default) = GPU if there is one (train_mnist.py:323-327).
    The MI355X decoder has no CPU path, so a CPU request is refused up front.  Under data-parallel execution the
    device is the rank's local one (dp.init_process_group: LOCAL_RANK, or 0 for every rank in the shared-GPU rehearsal)."""
    if d == -1 or not torch.cuda.is_available():
        raise SystemExit("spatial_vae_amd: the decoder runs on an MI355X only (-d -1 / no GPU is not supported)")
    idx = d if d >= 0 else local
    if world > 1:
        idx = local
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def make_run_dir(prefix, args):
    out = "outputs_{}".format(prefix)
    trained = os.path.join(out, "trained")
    os.makedirs(trained, exist_ok=True)
    os.makedirs(os.path.join(out, "images"), exist_ok=True)
    with open(os.path.join(out, "command.txt"), "w") as f:
        for k, v in sorted(vars(args).items()):
            print("{}: {}".format(k, v), file=f)
    return out, trained


def save_label(args):
    """src/misc_tools.py:15-28: '<prefix>_' + z<z_dim>pnl<p_num_layers>qnl<q_num_layers>nl<num_layers>ep<num_epochs> in
    the order the flags were declared."""
    names = {"z_dim": "z", "p_num_layers": "pnl", "q_num_layers": "qnl", "num_layers": "nl", "num_epochs": "ep"}
    label = args.save_prefix + "_"
    for k, v in vars(args).items():
        if k in names:
            label += names[k] + str(v)
    return label


def image_grid(images, nrow, padding=3, pad_value=0.5):
    """torchvision.utils.make_grid + the uint8 conversion of save_image (torchvision 0.8.2 is what the reference pins;
    it is not installed here, so this follows its published algorithm -- parity unpinned): images (B, C, h, w) in [0, 1]
    -> (H, W, 3) uint8.  Single-channel batches are replicated to RGB; cells are h+padding x w+padding on a
    pad_value canvas; value*255 + 0.5, clamped, truncated."""
    t = np.asarray(images, np.float32)
    if t.shape[1] == 1:
        t = np.repeat(t, 3, axis=1)
    B, C, h, w = t.shape
    if B == 1:
        grid = t[0]
    else:
        xmaps = min(nrow, B)
        ymaps = int(math.ceil(float(B) / xmaps))
        H, W = h + padding, w + padding
        grid = np.full((C, H * ymaps + padding, W * xmaps + padding), pad_value, np.float32)
        for k in range(B):
            r, c = divmod(k, xmaps)
            grid[:, r * H + padding:r * H + padding + h, c * W + padding:c * W + padding + w] = t[k]
    arr = np.clip(grid * np.float32(255) + np.float32(0.5), 0, 255).astype(np.uint8)
    return np.transpose(arr, (1, 2, 0))


def export_batch_as_image(data, output, image_dims):
    """src/misc_tools.py:30-39: a (B, N[, C]) batch as one PNG, int(sqrt(B)) images per row."""
    from PIL import Image
    B = data.size(0)
    images = data.detach().float().reshape(B, image_dims[0], image_dims[1], -1).permute(0, 3, 1, 2).cpu().numpy()
    Image.fromarray(image_grid(images, int(B ** 0.5))).save(output)


def save_models(path_prefix, epoch_str, p_net, q_net, device=None):
    """torch.save(<whole module>) under the reference's names (src/misc_tools.py:88-104).  The reference moves the live
    module (net.eval().cpu(), save, net.cuda()); here a detached CPU COPY is saved instead: the live parameters are
    views into dp.TrainStep's flat buffers, and a Module._apply round trip would re-allocate them -- the optimiser
    would keep stepping the orphaned flat buffer while the forward pass read stale module tensors."""
    for tag, net in (("generator", p_net), ("inference", q_net)):
        sinks = net.__dict__.pop("_grad_sinks", None)       # views into a training buffer: not part of the model
        try:
            snapshot = copy.deepcopy(net).eval().cpu()
        finally:
            if sinks is not None:
                net._grad_sinks = sinks
        torch.save(snapshot, "{}_{}_epoch{}.sav".format(path_prefix, tag, epoch_str))


def _metric_vector(out):
    """(elbo, log_p, kl) of an eval_minibatch result as one 3-vector (ops.elbo_head returns views of one)."""
    base = getattr(out[0], "_base", None)
    if base is not None and base.numel() == 3:
        return base.detach()
    return torch.stack([out[0].detach(), out[1].detach(), out[2].detach()])


def _take(t, sel, device):
    """Rows `sel` of a dataset tensor on `device`: a gather in HBM when the dataset is resident there (the default, as the
    reference preloads: train_mnist.py:329-332), a host gather + one asynchronous upload under --no-preload."""
    if t.device == device:
        return t[sel.to(device)]
    rows = t[sel.cpu()]
    return (rows.pin_memory() if device.type == "cuda" else rows).to(device, non_blocking=True)


def run_epoch(script, step, x, batches, train, N, epoch, num_epochs, rank, world, progress_every, extra):
    """One pass over `batches` (a list of index tensors into the resident data; every rank holds the same list).
    Training uses dp.TrainStep (forward, backward, all-reduce, Adam); evaluation only the forward (it is stochastic in
    the reference too: eval_model draws noise, train_mnist.py:174-226).

    Data parallel: rank g works on rows [lo, hi) of each GLOBAL minibatch.  The N(0,1) draw is made for the whole
    global minibatch from a generator every rank seeded identically (extra["noise_gen"]) and sliced like the data, and
    so are the augmentation angles, so a G-rank step computes what the 1-rank step computes (up to fp32 summation
    order).  Training metrics come back inside the gradient all-reduce (step.metrics); evaluation metrics are
    collected per batch and all-reduced ONCE per epoch.  A rank with an empty slice (ragged last batch smaller than the
    world) contributes zeros."""
    p_net, q_net = step.p_net, step.q_net
    p_net.train(train)
    q_net.train(train)
    data = extra["data"]
    mean = RunningMean(x.device)
    inf_dim = extra["inf_dim"]
    noise_list = extra.get("noise")
    pending = []
    for it, idx in enumerate(batches):
        gb = idx.numel()
        lo, hi = dp.shard_bounds(gb, rank, world)
        sel = idx[lo:hi]
        y = _take(data["y"], sel, x.device)
        args = (y,)
        if script == "particles":
            ctf = _take(data["ctf"], sel, x.device) if data.get("ctf") is not None else None
            args = (y, extra.get("mask"), ctf)
        kw = dict(extra.get("kw", {}))
        if noise_list is not None:
            r = noise_list[it]
        else:       # no prepared draws: the reference's per-minibatch draw from the global CPU generator (train_mnist.py:38)
            r = torch.empty(gb, inf_dim).normal_().to(x.device, non_blocking=True)
        kw["noise"] = r[lo:hi]
        out = None
        if train:
            kw.update(extra.get("train_kw", {}))        # augmentation applies to training steps only (train_galaxy.py:204)
            if world > 1 and kw.get("augment_rotation") and step.eval_kwargs.get("rotate"):
                kw["offset"] = E.draw_offsets(step.eval_kwargs["rotate"], gb)[lo:hi]   # np.random is seeded alike on all ranks
            out = step(x, *args, weight=(hi - lo) / gb, global_batch=gb, **kw)
            # the data-parallel metric tail (also the one-rank RCCL rehearsal, and any weight != 1) is overwritten by the
            # next step: keep a copy of it, not a reference
            mean.update(gb, step.metrics, volatile=step.metrics is step.grads.tail)
        else:
            vals = torch.zeros(3, device=x.device)
            if hi > lo:
                with torch.no_grad():
                    call = dict(step.eval_kwargs)
                    call.update(kw)
                    out = step.eval_minibatch(x, *args, p_net, q_net, **call)
                vals = _metric_vector(out)
            if world > 1:
                pending.append((gb, vals * ((hi - lo) / gb)))
            else:
                mean.update(gb, vals)
        if it == 0 and extra.get("dump") and rank == 0 and out is not None:   # first batch of a save-interval epoch (train_mnist.py:214-224)
            extra["dump"](y, out[3] if len(out) > 3 else None)
        if train and rank == 0 and progress_every > 0 and (it + 1) % progress_every == 0:
            e, g, k = mean.values()
            print("# [{}/{}] training {:.1%}, ELBO={:.5f}, Error={:.5f}, KL={:.5f}".format(
                epoch + 1, num_epochs, mean.seen / N, e, g, k), end="\r", file=sys.stderr)
    if pending:
        allv = torch.stack([v for _, v in pending])
        torch.distributed.all_reduce(allv)
        for (gb, _), v in zip(pending, allv):
            mean.update(gb, v)
    if train and rank == 0 and progress_every > 0:
        print(" " * 80, end="\r", file=sys.stderr)
    return mean.values()


def train_main(script, args, build):
    """`build(args, device)` returns dict(y_train, y_test, ctf_train, ctf_test, mask, n, m, channels,
    p_net, q_net, rotate, translate, table)."""
    rank, world, local = dp.init_process_group(device_is_gpu=True)
    device = pick_device(args.device, world, local)
    # Randomness is consumed from torch's GLOBAL CPU generator and np.random in the order the reference's main() consumes
    # them: default initialisation of p_net then q_net, one draw for the sample-image pass over the validation
    # loader, then per epoch the two draws of iter(DataLoader(shuffle=True)), one N(0,1) draw per training minibatch, one draw
    # for the validation loader and one N(0,1) draw per validation minibatch (plus the display helpers' on dump epochs).  So
    # torch.manual_seed(s) in front of this function -- or --seed s, an addition: the reference has no such flag -- follows the
    # trajectory of the reference's CPU path under the same seed.  Under data-parallel execution every rank seeds alike
    # (--seed, else rank 0's seed), makes the same draws and slices [lo:hi) of each global minibatch; np.random (the dataset
    # shuffle of train_galaxy.py:372 inside build(), the augmentation angles) is seeded alike too.
    seed = getattr(args, "seed", None)
    if seed is None and world > 1:
        seed = dp.shared_seed(device)
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed % (2 ** 32))
    if getattr(args, "gemm", None):                 # before the first decoder call: buffer sizes depend on the mode
        from . import _lib
        _lib.set_gemm_mode(args.gemm)
    start = time.time()
    prefix = args.save_prefix
    out_dir = trained = None
    if script != "particles":
        if prefix is None:
            raise SystemExit("--save_prefix is required (the reference crashes without it: src/misc_tools.py:22)")
        if rank == 0:
            out_dir, trained = make_run_dir(prefix, args)
    cfg = build(args, device)
    p_net, q_net = cfg["p_net"].to(device), cfg["q_net"].to(device)
    if rank == 0 and out_dir:
        with open(os.path.join(out_dir, "models.txt"), "w") as f:
            print(p_net, file=f)
            print(q_net, file=f)
    x = coord_grid(cfg["n"], cfg["m"]).to(device)
    fn = {"mnist": E.eval_minibatch_mnist, "galaxy": E.eval_minibatch_galaxy, "particles": E.eval_minibatch_particles}[script]
    step = dp.TrainStep(p_net, q_net, fn, lr=args.learning_rate, rotate=cfg["rotate"], translate=cfg["translate"],
                        dx_scale=args.dx_scale, theta_prior=args.theta_prior)
    print("# using priors: theta={}, dx={}".format(args.theta_prior, args.dx_scale), file=sys.stderr)
    num_epochs = args.num_epochs
    digits = int(math.log10(num_epochs)) + 1
    bs = args.minibatch_size
    # the dataset is resident in HBM (the reference preloads it too: train_mnist.py:329-332) unless --no-preload
    # (train_particles.py:317, :405-413) keeps it in host memory, from where each minibatch is gathered and uploaded
    home = torch.device("cpu") if getattr(args, "no_preload", False) else device
    tr = {"y": cfg["y_train"].to(home), "ctf": None if cfg.get("ctf_train") is None else cfg["ctf_train"].to(home)}
    te = {"y": cfg["y_test"].to(home), "ctf": None if cfg.get("ctf_test") is None else cfg["ctf_test"].to(home)}
    mask = cfg.get("mask")
    mask = mask.to(device) if mask is not None else None
    N = tr["y"].size(0)
    if world > 1:       # every rank must hold the SAME dataset: the ranks slice one global minibatch by index
        dp.assert_same_on_all_ranks(torch.stack([tr["y"].double().sum(), te["y"].double().sum()]).to(device), "the dataset")
    inf_dim = q_net.latent_dim
    out = sys.stdout
    header = cfg["table"]
    if rank == 0:
        print("\t".join(header), file=out)
    train_lines, val_lines = ["\t".join(header)], ["\t".join(header)]
    z_delay = getattr(args, "z_delay", 0)
    label = save_label(args) if out_dir else None
    ntest = te["y"].size(0)
    if script != "particles":                                        # MiscTools.sample_images: one pass is started over the
        loader_order(ntest, False)                                   # validation loader (train_mnist.py:402) -> one draw
        if out_dir:
            export_batch_as_image(te["y"][:bs], "{}/images/_sample_{}.png".format(out_dir, label), [cfg["n"], cfg["m"]])
    for epoch in range(num_epochs):
        kw = {}
        if script != "mnist":
            kw["z_scale"] = 0 if epoch < z_delay else 1
        batches, noise = train_pass_plan(N, bs, inf_dim, device, home)    # same order and draws on every rank
        train_kw = {"augment_rotation": True} if cfg.get("augment") and script != "mnist" else {}
        t_epoch = time.time()
        e, g, k = run_epoch(script, step, x, batches, True, N, epoch, num_epochs, rank, world, args.progress_every,
                            dict(data=tr, mask=mask, kw=kw, train_kw=train_kw, inf_dim=inf_dim, noise=noise))
        if rank == 0:       # run_epoch's values() synchronised: the epoch's training pass is complete
            print("# epoch {}: {} training images in {:.3f} s = {:.0f} images/s".format(
                epoch + 1, N, time.time() - t_epoch, N / max(time.time() - t_epoch, 1e-9)), file=sys.stderr)
        dump = None
        shapes = None
        if script != "particles" and (epoch + 1) % args.save_interval == 0:
            # the display helpers' draws are made on every rank (one random stream), the files written by rank 0
            shapes = display_draw_shapes(script, inf_dim, args.z_dim)
        tb, noise, shown = eval_pass_plan(ntest, bs, inf_dim, device, home, shapes)
        if shapes and out_dir:
            dump = _image_dumper(script, step, x, cfg, out_dir, str(epoch + 1).zfill(digits), label, kw, args.z_dim, shown)
        ev = run_epoch(script, step, x, tb, False, ntest, epoch, num_epochs, rank, world, 0,
                       dict(data=te, mask=mask, kw=kw, dump=dump, inf_dim=inf_dim, noise=noise))
        if rank == 0:
            if script == "particles":
                print("\t".join([str(epoch + 1), "train", str(e), str(g), str(k)]), file=out)
                print("\t".join([str(epoch + 1), "test", str(ev[0]), str(ev[1]), str(ev[2])]), file=out)
            else:
                line = "\t".join(map(str, [epoch, e, g, k]))
                train_lines.append(line)
                print(line, file=out)
                line = "\t".join(map(str, [epoch, ev[0], ev[1], ev[2]]))
                val_lines.append(line)
                print(line, file=out)
            out.flush()
            if script == "particles" and prefix is not None and (epoch + 1) % args.save_interval == 0:
                save_models(prefix, str(epoch + 1).zfill(digits), p_net, q_net, device)
    if rank == 0 and script != "particles":
        save_models(os.path.join(trained, prefix), str(num_epochs).zfill(digits), p_net, q_net, device)
        with open(os.path.join(out_dir, "train.txt"), "w") as f:
            print("\n".join(train_lines), file=f)
        with open(os.path.join(out_dir, "val.txt"), "w") as f:
            print("\n".join(val_lines), file=f)
        print("Elapsed time: {:.1f} s".format(time.time() - start))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def display_draw_shapes(script, inf_dim, z_dim):
    """Shapes of the N(0,1) draws eval_model's image dump makes for a first minibatch of B images: minibatch_for_display
    draws (B, inf_dim) (train_mnist.py:107, train_galaxy.py:146); galaxy's random_minibatch_generator then (B, z_dim)
    (train_galaxy.py:177)."""
    if script == "galaxy":
        return lambda B: [(B, inf_dim), (B, z_dim)]
    return lambda B: [(B, inf_dim)]


def _image_dumper(script, step, x, cfg, out_dir, epoch_str, label, kw, z_dim, shown=(None, None)):
    """The PNG dumps of eval_model (train_mnist.py:214-224, train_galaxy.py:275-292): <epoch>_dis_ = decoded from the
    content latents on the unposed grid, <epoch>_ = the posed reconstruction y_hat of the same batch, galaxy also
    <epoch>_rnd_ = decoded prior samples.  `shown`: the helpers' N(0,1) draws, made by pass_noise in the reference's order."""
    dims = [cfg["n"], cfg["m"]]
    p_net, q_net = step.p_net, step.q_net
    shown = list(shown) + [None, None]

    def dump(y, y_hat):
        base = "{}/images/{}".format(out_dir, epoch_str)
        rows = y.size(0)                        # data parallel: rank 0's slice [0, rows) of the first global minibatch
        shown[0] = shown[0][:rows] if shown[0] is not None else None
        shown[1] = shown[1][:rows] if shown[1] is not None else None
        if script == "mnist":
            dis = E.minibatch_for_display(x, y, p_net, q_net, rotate=cfg["rotate"], translate=cfg["translate"], noise=shown[0])
        else:
            zs = kw.get("z_scale", 1)
            dis = E.minibatch_for_display_galaxy(x, y, q_net, p_net, rotate=cfg["rotate"], translate=cfg["translate"], z_scale=zs,
                                                 noise=shown[0])
            rnd = E.random_minibatch_generator(x, y, p_net, z_dim, z_scale=zs, noise=shown[1])
            export_batch_as_image(rnd, "{}_rnd_{}.png".format(base, label), dims)
        export_batch_as_image(dis, "{}_dis_{}.png".format(base, label), dims)
        if y_hat is not None:
            export_batch_as_image(y_hat, "{}_{}.png".format(base, label), dims)

    return dump


def synthetic_images(kind, count, n, m, channels, seed):
    rs = np.random.RandomState(seed)
    if kind == "particles":
        return rs.normal(size=(count, n, m)).astype(np.float32)
    shape = (count, n, m) if channels == 1 else (count, n, m, channels)
    u = rs.uniform(size=shape)
    keep = rs.uniform(size=shape) > (0.8 if channels == 1 else 0.0)
    return np.floor(u * keep * 255.0).astype(np.float32)
