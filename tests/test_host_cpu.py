"""CPU-only checks of the host side: the C-ABI library loads and exports what include/svae.h declares,
argument/descriptor validation, workspace sizing, the reference-API mirror (state-dict names, the
no-CPU-fallback rule) and the data-parallel helper on gloo with world_size 2.  No GPU compute here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from spatial_vae_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol():
    L = _lib()
    lib = L.lib()
    header = open(os.path.join(ROOT, "include", "svae.h")).read()
    declared = set(re.findall(r"\b(svae_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed from include/svae.h"
    for name in declared:
        assert hasattr(lib, name), "libsvae_hip.so does not export %s" % name
    assert set(L.EXPORTS) == set(L.declared_in_header()) == declared      # the binding's static list IS the header's
    assert lib.svae_abi_version() == 2


def _desc(**kw):
    L = _lib()
    d = L.Desc()
    base = dict(B=4, N=49, H=20, L=2, Zd=2, C=1, in_dim=2, act=0, flags=0)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_workspace_and_saved_sizes():
    lib = _lib().lib()
    d = _desc()
    saved = lib.svae_saved_bytes(ctypes.byref(d))
    ws = lib.svae_workspace_bytes(ctypes.byref(d))
    # Npad = 64, Mp = 256, Hp = 32: two saved activation planes of Mp*Hp floats
    assert saved >= 2 * 256 * 32 * 4 and saved % 256 == 0
    assert ws > 2 * 256 * 32 * 4 and ws % 256 == 0
    big = _desc(B=256, N=784, H=500)
    # BASELINE cfg 2: 2 activation planes of 419 MB + the packed weights (2 x 1 MB), tables (4 MB) and poses that ride along
    extra = lib.svae_saved_bytes(ctypes.byref(big)) - 2 * 204800 * 512 * 4
    assert 2 * 512 * 512 * 4 + 256 * 512 * 8 * 4 + 256 * 16 <= extra <= 2 * 512 * 512 * 4 + 256 * 512 * 8 * 4 + 256 * 16 + 2048


@pytest.mark.parametrize("bad", [dict(B=0), dict(L=0), dict(L=9), dict(C=5), dict(in_dim=3), dict(act=7),
                                 dict(Zd=0, flags=2)])
def test_invalid_descriptors_are_rejected(bad):
    lib = _lib().lib()
    d = _desc(**bad)
    assert lib.svae_workspace_bytes(ctypes.byref(d)) == 0
    assert lib.svae_last_error()          # a message was recorded


def test_forward_rejects_null_pointers_without_touching_the_gpu():
    L = _lib()
    lib = L.lib()
    d = _desc()
    p, pose = L.Params(), L.Pose()
    rc = lib.svae_decoder_forward(ctypes.byref(d), ctypes.byref(p), ctypes.byref(pose), None, None, None, None, None, 0, None)
    assert rc == -1 and b"parameter" in lib.svae_last_error()
    assert lib.svae_gaussian_loglik(1, 4, 3, None, None, None, None, 0, None, None, None, 0, None) == -1


def test_model_mirror_matches_reference_state_dict_names():
    import spatial_vae.models as models
    for name in ("mnist_rt", "mnist_L3_resid", "mnist_bilinear_expand", "mnist_z0", "galaxy_rgb"):
        case = C.CASES_BY_NAME[name]
        inp = C.build_inputs(case)
        act = {"tanh": nn.Tanh, "leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "sigmoid": nn.Sigmoid}[case["act"]]
        p = models.SpatialGenerator(case["z_dim"], case["H"], n_out=case["n_out"], num_layers=case["L"], activation=act,
                                    softplus=case["softplus"], resid=case["resid"], expand_coords=case["expand_coords"],
                                    bilinear=case["bilinear"])
        assert list(p.state_dict().keys()) == list(inp["p_state"].keys())
        p.load_state_dict({k: torch.from_numpy(v) for k, v in inp["p_state"].items()})
        n_in = case["n"] * case["m"] * (case["n_out"] if case["script"] == "galaxy" else 1)
        q = models.InferenceNetwork(n_in, C.inf_dim(case), case["q_hidden"], num_layers=case["q_layers"], activation=act,
                                    resid=case["resid"])
        assert list(q.state_dict().keys()) == list(inp["q_state"].keys())


def test_default_init_consumes_rng_like_the_reference():
    """p_net then q_net under one seed must give the tensors plain nn.Linear construction gives in the
    reference's attribute order: coord_linear, latent_linear, layers..."""
    import spatial_vae.models as models
    torch.manual_seed(0)
    p = models.SpatialGenerator(2, 8, num_layers=2)
    torch.manual_seed(0)
    ref = [nn.Linear(2, 8), nn.Linear(2, 8, bias=False), nn.Linear(8, 8), nn.Linear(8, 1)]
    got = [p.coord_linear, p.latent_linear, p.layers[1], p.layers[3]]
    for a, b in zip(got, ref):
        assert torch.equal(a.weight, b.weight)


def test_there_is_no_cpu_fallback():
    import spatial_vae.models as models
    p = models.SpatialGenerator(2, 8, num_layers=2)
    with pytest.raises(RuntimeError, match="HIP device"):
        p(torch.zeros(2, 5, 2), torch.zeros(2, 2))
    from spatial_vae_amd import ops
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.bce_loglik(torch.rand(2, 5), torch.rand(2, 5))


def test_unsupported_activation_is_refused():
    import spatial_vae.models as models
    with pytest.raises(NotImplementedError):
        models.SpatialGenerator(2, 8, activation=nn.ELU)


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product packages or the CLIs may reference it."""
    offenders = []
    for sub in ("spatial_vae_amd", "spatial_vae"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".h", ".hip")):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt and f.endswith(".py"):
                        offenders.append(os.path.join(sub, f))
    assert not offenders, offenders


def test_shard_bounds_cover_the_batch():
    from spatial_vae_amd import dp
    for n in (1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["SVAE_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from spatial_vae_amd import dp
rank, world, _ = dp.init_process_group(device_is_gpu=False)
torch.manual_seed(0)
net = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 1))
flat = dp.FlatGrads(list(net.parameters()))
x = torch.arange(7 * 6, dtype=torch.float32).reshape(7, 6) / 10.0
lo, hi = dp.shard_bounds(7, rank, world)            # ragged: 4 + 3 rows
loss = net(x[lo:hi]).pow(2).mean()
loss.backward()
flat.all_reduce(weight=(hi - lo) / 7.0)
# single-process reference on the whole batch
torch.manual_seed(0)
ref = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 1))
ref(x).pow(2).mean().backward()
g_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
packed = torch.cat([flat.flat[o:o + p.numel()] for p, o in zip(flat.params, flat.offsets)])   # without the alignment gaps
err = (packed - g_ref).abs().max().item() / g_ref.abs().max().item()
assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in net.parameters())
print("rank", rank, "err", err)
assert err < 1e-6, err
dist.destroy_process_group()
'''


def test_data_parallel_allreduce_gloo_world2(tmp_path):
    """Two CPU ranks with a ragged split reproduce the single-process gradient of the global mean."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, SVAE_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("err") == 2


def test_gemm_mode_switch_is_validated_and_needs_no_gpu(monkeypatch):
    """svae_gemm_mode_set / _get (include/svae.h): default from SVAE_GEMM, explicit set, bad values refused."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); from spatial_vae_amd import _lib; L = _lib.lib();"
            "assert _lib.gemm_mode() == os.environ.get('WANT'); _lib.set_gemm_mode('fp16x3'); assert _lib.gemm_mode() == 'fp16x3';"
            "_lib.set_gemm_mode('fp32'); assert _lib.gemm_mode() == 'fp32'; rc = L.svae_gemm_mode_set(7); assert rc != 0;"
            "assert b'unknown mode' in L.svae_last_error(); print('ok')") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_val, want in ((None, "fp32"), ("fp16x3", "fp16x3"), ("nonsense", "fp32")):
        env = dict(os.environ, WANT=want)
        env.pop("SVAE_GEMM", None)
        if env_val is not None:
            env["SVAE_GEMM"] = env_val
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]
