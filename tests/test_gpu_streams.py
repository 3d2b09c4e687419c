"""The boundary is re-entrant per stream (SURVEY.md 8(b) "Threading / streams"): every entry point launches on the hipStream_t it is handed
(torch's CURRENT stream through the Python layer) and keeps no state between calls beyond per-device atomics.  BASELINE config #2's batch
on a fresh stream, on two streams at once (one host thread, then one host thread per stream: ctypes drops the GIL for the call), one
training step on a side stream -- each BIT-identical to the default-stream result -- and, in a fresh process, the FIRST call of every
kernel family made from a non-default stream (the lazy 160 KB-LDS opt-in and CU-count lookups happen there)."""
import json
import os
import subprocess
import sys
import threading
from types import SimpleNamespace

import pytest
import torch

from nerf_pytorch_paeng_amd import nerf_process as NP
from nerf_pytorch_paeng_amd import ops, synthetic, weights

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"fp32": {}, "bf16": {"bf16": True}, "f16s": {"f16s": True}}
KEYS = ("rgb_c", "disp_c", "rgb_f", "disp_f")


def make_opts():
    return SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0)


@pytest.fixture(scope="module")
def packed():
    return weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), DEV)


@pytest.fixture(scope="module")
def rays():
    K, H, W = synthetic.lego_camera()
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, 4096, 0)).to(DEV)
    o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
    return torch.cat([o, d], -1).contiguous()


def render(rays, packed, mode):
    out = NP.render_rays(rays, packed, None, make_opts(), seed=7, **MODES[mode])
    return tuple(out[k] for k in KEYS)


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("mode", list(MODES))
def test_fresh_stream_and_two_streams_equal_the_default_stream(packed, rays, mode):
    want = render(rays, packed, mode)
    torch.cuda.synchronize(DEV)
    # 1. a fresh side stream (its own workspace and outputs: allocated under the stream's context)
    s = torch.cuda.Stream(DEV)
    s.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(s):
        got = render(rays, packed, mode)
    s.synchronize()
    assert same(got, want)
    # 2. two streams, enqueued alternately from this thread, three rounds each: six launch sequences in flight over two queues
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    res = {s1: [], s2: []}
    for _ in range(3):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                res[st].append(render(rays, packed, mode))
    s1.synchronize(); s2.synchronize()
    assert all(same(r, want) for st in res for r in res[st])
    # 3. the default stream is still what it was
    assert same(render(rays, packed, mode), want)


def test_two_host_threads_two_streams(packed, rays):
    """One host thread per stream, all three precision modes in each, started together: the library is entered concurrently."""
    want = {m: render(rays, packed, m) for m in MODES}
    torch.cuda.synchronize(DEV)
    out, errs = {}, []
    gate = threading.Barrier(2)

    def work(idx):
        try:
            torch.cuda.set_device(DEV)
            st = torch.cuda.Stream(DEV)
            gate.wait()
            with torch.cuda.stream(st):
                got = [(m, render(rays, packed, m)) for _ in range(2) for m in (list(MODES) if idx == 0 else list(MODES)[::-1])]
            st.synchronize()
            out[idx] = got
        except Exception as e:                                  # noqa: BLE001 -- reported by the main thread
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert all(same(g, want[m]) for i in out for m, g in out[i])


def _train_step(rays, f16s, stream=None):
    from nerf_pytorch_paeng_amd import train_path
    from nerf_pytorch_paeng_amd.model import NeRF
    sd = synthetic.make_state_dict(3, 8, 256)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(DEV))
    with ctx:
        model = NeRF(8, 256, 63, 27).to(DEV)
        model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
        opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0)
        tgt = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(5)).to(DEV)
        out = train_path.render_train(rays, model, opts, seed=9, f16s=f16s)
        (torch.mean((out["rgb_c"] - tgt) ** 2) + torch.mean((out["rgb_f"] - tgt) ** 2)).backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        rgb = out["rgb_f"].detach().clone()
    return rgb, grads


@pytest.mark.parametrize("f16s", [False, True])
def test_training_step_on_a_side_stream(rays, f16s):
    """Forward with stash + composite backward + backward-data + backward-weights + the slice reductions, all on a side stream: the colours
    and the flat gradient equal the default-stream step's bit for bit (the reductions are ordered, there are no float atomics)."""
    r = rays[:1024].contiguous()
    rgb0, g0 = _train_step(r, f16s)
    torch.cuda.synchronize(DEV)
    s = torch.cuda.Stream(DEV)
    s.wait_stream(torch.cuda.current_stream(DEV))
    rgb1, g1 = _train_step(r, f16s, s)
    s.synchronize()
    assert torch.equal(rgb0, rgb1)
    assert torch.equal(g0, g1), float((g0 - g1).abs().max())


CHILD = r"""
import json, sys, zlib
sys.path.insert(0, {root!r})
import torch
from types import SimpleNamespace
from nerf_pytorch_paeng_amd import nerf_process as NP, ops, synthetic, weights
dev = torch.device("cuda:0")
s = torch.cuda.Stream(dev)                      # NOTHING of the library has run on the default stream in this process
def crc(t):
    return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes())
res = {{}}
with torch.cuda.stream(s):
    packed = weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), dev)
    K, H, W = synthetic.lego_camera()
    pix = torch.from_numpy(synthetic.pixel_batch(H, W, 4096, 0)).to(dev)
    o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
    rays = torch.cat([o, d], -1).contiguous()
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                           data_type="blender", gpu_ids=[0], rank=0)
    for mode, kw in (("fp32", {{}}), ("bf16", {{"bf16": True}}), ("f16s", {{"f16s": True}})):
        out = NP.render_rays(rays, packed, None, opts, seed=7, **kw)
        s.synchronize()
        res[mode] = [crc(out[k]) for k in ("rgb_c", "disp_c", "rgb_f", "disp_f")]
    sys.path.insert(0, {tests!r})
    from test_gpu_streams import _train_step
    for f16s in (False, True):
        rgb, g = _train_step(rays[:1024].contiguous(), f16s, s)
        s.synchronize()
        res["train_f16s" if f16s else "train"] = [crc(rgb), crc(g)]
print("RESULT " + json.dumps(res))
"""


def test_first_calls_of_a_fresh_process_on_a_side_stream(packed, rays):
    """A fresh process whose FIRST call of each kernel family (fp32 / bf16 / f16s render, training forward + backward, both modes) is made
    on a non-default stream: the lazy per-device state (LDS opt-in per kernel, CU count) is filled from there.  Its results (CRC32 of the
    output bytes) equal this process's default-stream results."""
    import zlib

    def crc(t):
        return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes())
    want = {m: [crc(t) for t in render(rays, packed, m)] for m in MODES}
    for f16s in (False, True):
        rgb, g = _train_step(rays[:1024].contiguous(), f16s)
        want["train_f16s" if f16s else "train"] = [crc(rgb), crc(g)]
    torch.cuda.synchronize(DEV)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    got = json.loads(line[len("RESULT "):])
    assert got == want, (got, want)
