"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden fixtures.  Tolerances are stated per stage; fp32 end-to-end bar: per-pixel RGB within 1e-4
stage-wise (BASELINE.json north_star), with the sample_pdf branch-flip outlier fraction reported."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import nerf_process as NP
from nerf_pytorch_paeng_amd import ops, rays as RAYS, synthetic, weights
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder
from oracle import restate as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
T = torch.from_numpy


def g2d(a):
    return (T(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a).to(DEV)


def err(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, err_msg=what)


def make_opts(**kw):
    base = dict(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288,
                data_type="blender", gpu_ids=[0], rank=0)
    base.update(kw)
    return SimpleNamespace(**base)


@pytest.fixture(scope="module")
def packed_big():
    return weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), DEV)


@pytest.fixture(scope="module")
def lego_rays():
    K, H, W = synthetic.lego_camera()
    pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
    pix = T(synthetic.pixel_batch(H, W, 4096, 0)).to(DEV)
    o, d = ops.make_o_d_pixels(W, H, K, pose, pix)
    return torch.cat([o, d], -1).contiguous()


# ---------------------------------------------------------------------------------------------------
def test_mfma_fragment_layout_selftest():
    ops.selftest_mfma(DEV)


def test_raygen_F1(golden):
    g = golden("F1_raygen")
    for key in ("lego0", "lego1", "fern0", "fern1"):
        H, W = (int(v) for v in g[f"{key}_HW"])
        o, d = RAYS.make_o_d(W, H, g[f"{key}_K"], g2d(g[f"{key}_pose"][:3, :4]))
        assert o.shape == d.shape == (H, W, 3) and o.stride()[:2] == (0, 0)          # stride-0 origin view, rays.py:33
        ys, xs = g[f"{key}_ys"], g[f"{key}_xs"]
        close(o[ys, xs], g[f"{key}_o"], 0, what=key)
        close(d[ys, xs], g[f"{key}_d"], 2e-7, 2e-7, what=key)                        # 3-term dot: <= 1 ulp-ish
        on, dn = RAYS.get_rays_np(H, W, g[f"{key}_K"], g[f"{key}_pose"][:3, :4])
        close(dn[ys, xs], g[f"{key}_d"], 2e-7, 2e-7)
        # whole image against the oracle, and the row-window form used for multi-GPU sharding
        oo, od = R.make_o_d(W, H, g[f"{key}_K"], T(g[f"{key}_pose"][:3, :4]))
        close(d, od, 2e-7, 2e-7)
        _, dwin = ops.make_o_d(W, H, g[f"{key}_K"], g[f"{key}_pose"], DEV, row0=H // 3, n_rows=5, want_origins=False)
        assert torch.equal(dwin, d[H // 3:H // 3 + 5])
    o, d = RAYS.make_o_d(4, 3, g["kat_K"], g2d(g["kat_pose"]))
    close(d, g["kat_d"], 0); close(o, g["kat_o"], 0)


def test_raygen_pixels_matches_full(lego_rays):
    K, H, W = synthetic.lego_camera()
    pose = synthetic.pose_spherical(0.0, -30.0, 4.0)
    _, d = ops.make_o_d(W, H, K, pose, DEV, want_origins=False)
    pix = T(synthetic.pixel_batch(H, W, 4096, 0)).to(DEV)
    assert torch.equal(lego_rays[:, 3:], d.reshape(-1, 3)[pix])


def test_ndc_F2(golden):
    g = golden("F2_ndc")
    o, d = NP.ndc_rays(int(g["H"]), int(g["W"]), float(g["focal"]), float(g["near"]), g2d(g["o_in"]), g2d(g["d_in"]))
    close(o, g["o_out"], 1e-6, 1e-6); close(d, g["d_out"], 1e-6, 1e-6)
    # stride-0 origin (make_o_d's expanded view) is accepted
    o1 = g2d(g["o_in"][:1]).expand(256, 3)
    o2, _ = NP.ndc_rays(int(g["H"]), int(g["W"]), float(g["focal"]), 1.0, o1, g2d(g["d_in"]))
    oo, _ = R.ndc_rays(int(g["H"]), int(g["W"]), float(g["focal"]), 1.0, T(g["o_in"][:1]).expand(256, 3), T(g["d_in"]))
    close(o2, oo, 1e-6, 1e-6)


def test_fill_uniform_bit_exact_and_shard_invariant():
    a = ops.fill_uniform(7, 0, 0, 5000, 64, DEV).cpu().numpy()
    assert np.array_equal(a, R.counter_uniform(7, 0, 0, 5000, 64))
    b = ops.fill_uniform(7, 0, 1234, 100, 64, DEV).cpu().numpy()
    assert np.array_equal(b, a[1234:1334])
    c = ops.fill_uniform(7, 1, 0, 16, 128, DEV).cpu().numpy()
    assert np.array_equal(c, R.counter_uniform(7, 1, 0, 16, 128))


def test_stratified_F3(golden):
    g = golden("F3_stratified")
    z = ops.stratified_z(float(g["near"]), float(g["far"]), g2d(g["t_rand"]))
    close(z, g["z_vals"], 5e-7)
    for S, near, far in ((64, 0.0, 1.0), (32, 2.0, 6.0), (1, 2.0, 6.0), (7, 0.5, 9.0)):
        t = T(R.counter_uniform(1, 0, 0, 33, S))
        close(ops.stratified_z(near, far, t.to(DEV)), R.stratified_z(33, near, far, S, t), 1e-6)


def _flip_fraction(got, want, tol):
    bad = (got.cpu() - want).abs() > tol
    return float(bad.float().mean())


def test_sample_pdf_F4(golden):
    g = golden("F4_sample_pdf")
    bins, w, u = g2d(g["bins"]), g2d(g["weights"]), g2d(g["u"])
    opts = make_opts()
    s_det = NP.sample_pdf(bins, w, 128, det=True, opts=opts)
    s_rnd = NP.sample_pdf(bins, w, 128, det=False, opts=opts, u=u)
    # sample_pdf is discontinuous in the cdf (searchsorted bin flips, denom<1e-5 branch): almost all samples
    # agree to 2e-6; count the flips instead of hiding them
    for got, key in ((s_det, "samples_det"), (s_rnd, "samples_rand")):
        frac = _flip_fraction(got, T(g[key]), 5e-6)
        print(f'sample_pdf {key}: samples off by >5e-6: {frac:.2e}')
        assert frac <= 5e-3, (key, frac)                 # observed 2.4e-3 .. 2.9e-3
        assert float((got.cpu() - T(g[key])).abs().median()) <= 5e-7
    kat = NP.sample_pdf(torch.linspace(2, 6, 5)[None].to(DEV), torch.tensor([[0.1, 0.0, 0.6, 0.3]], device=DEV), 6, det=True, opts=opts)
    close(kat, g["kat_samples"], 2e-6)
    # fine branch: merge + sort
    wc = torch.cat([torch.zeros(48, 1), T(g["weights"]), torch.zeros(48, 1)], -1).to(DEV)
    zf, zs = ops.fine_z(g2d(g["z_coarse"]), wc, 128, False, u, want_samples=True)
    assert torch.all(zf[:, 1:] >= zf[:, :-1])
    assert torch.equal(torch.sort(torch.cat([g2d(g["z_coarse"]), zs], -1), -1)[0], zf)       # exact multiset, sorted
    ff = _flip_fraction(zf, T(g["z_fine"]), 5e-6)
    assert ff <= 5e-3, ff                                # observed 2.7e-3


@pytest.mark.parametrize("n,Sc,Nf", [(37, 64, 128), (5, 3, 1), (9, 17, 40), (4, 64, 192), (3, 200, 821), (2, 1024, 1024)])
def test_fine_z_sort_is_torch_sort(n, Sc, Nf):
    """sort(cat(z_c, samples)) (nerf_process.py:67) for sample counts that are not powers of two (the sort network pads), with
    repeated depths, and with NaN weights (a diverged network): NaNs last like torch.sort, every slot written."""
    g = torch.Generator().manual_seed(n * Sc + Nf)
    z_c = torch.sort(2.0 + 4.0 * torch.rand(n, Sc, generator=g), -1)[0]
    z_c[:, Sc // 2] = z_c[:, Sc // 2 - 1]                                  # a tie
    w = torch.rand(n, Sc, generator=g)
    u = torch.rand(n, Nf, generator=g)
    zf, zs = ops.fine_z(z_c.to(DEV), w.to(DEV), Nf, False, u.to(DEV), want_samples=True)
    assert torch.equal(torch.sort(torch.cat([z_c.to(DEV), zs], -1), -1)[0], zf)
    want_s, _ = R.fine_z(z_c, w, Nf, False, u)
    if zf.numel() >= 1000:
        assert _flip_fraction(zf, want_s, 5e-6) <= 2e-2
    if n > 1:
        w[0, 1] = float("nan")                                              # ray 0: every sample NaN, the coarse depths are not
        zf = ops.fine_z(z_c.to(DEV), w.to(DEV), Nf, False, u.to(DEV))
        assert torch.equal(zf[0, :Sc], z_c[0].to(DEV)) and bool(torch.isnan(zf[0, Sc:]).all())
        assert not bool(torch.isnan(zf[1:]).any())


@pytest.mark.parametrize("Sc,Nf", [(512, 512), (1000, 24), (341, 683), (3, 1021)])
def test_largest_sample_counts_fused_equals_staged(Sc, Nf):
    """mi_nerf_render_rays composites, resamples and merges in ONE launch (composite_fine_z_kernel), whose LDS need (3 Sc - 2 + pow2(Sc + Nf)
    floats per ray) is larger than the staged fine_z kernel's.  The fine compositing takes at most 1024 depths, and within Sc + Nf <= 1024
    the fused bound always holds: the largest sample counts render, bit-identical to the staged sequence of entry points."""
    n = 6
    sd = synthetic.make_state_dict(2, 4, 128, skips=())
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    K, H, W = synthetic.lego_camera()
    pix = T(synthetic.pixel_batch(H, W, n, 3)).to(DEV)
    o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(10.0, -30.0, 4.0), pix)
    rays = torch.cat([o, d], -1).contiguous()
    t_rand, u = T(R.counter_uniform(1, 0, 0, n, Sc)).to(DEV), T(R.counter_uniform(1, 1, 0, n, Nf)).to(DEV)
    out = NP.render_rays(rays, packed, None, make_opts(N_samples_c=Sc, N_samples_f=Nf), t_rand=t_rand, u=u)
    z_c = ops.stratified_z(2.0, 6.0, t_rand)
    rgb_c, disp_c, _, w_c, _ = ops.composite(ops.mlp_rays(packed.net, packed.coarse, rays, z_c), z_c, rays, want_all=True)
    z_f = ops.fine_z(z_c, w_c, Nf, False, u)
    rgb_f, disp_f, *_ = ops.composite(ops.mlp_rays(packed.net, packed.fine, rays, z_f), z_f, rays)
    assert torch.equal(out["rgb_c"], rgb_c) and torch.equal(out["disp_c"], disp_c)
    assert torch.equal(out["rgb_f"], rgb_f) and torch.equal(out["disp_f"], disp_f) and torch.isfinite(rgb_f).all()
    with pytest.raises(Exception):                                             # one depth more than the compositing kernel takes
        NP.render_rays(rays, packed, None, make_opts(N_samples_c=Sc, N_samples_f=Nf + 1), t_rand=t_rand)


def test_posenc_embed_F5(golden):
    g = golden("F5_posenc")
    f10, d10 = get_positional_encoder(10)
    f4, d4 = get_positional_encoder(4)
    assert (d10, d4) == (63, 27) and f10.L == 10
    close(f10(g2d(g["pts"])), g["enc10"], 3e-7, what="gamma10")       # accurate range-reduced sin/cos (|arg| to 3e3)
    close(f4(g2d(g["pts"])), g["enc4"], 3e-7, what="gamma4")
    emb = ops.embed(g2d(g["rays8"]), g2d(g["z8"]), 10, 4)
    close(emb, g["embedded8"], 2e-4, what="embedded")                 # 2^9 * ulp(point) phase noise on the top band
    close(emb[:, :33], g["embedded8"][:, :33], 5e-6)
    big = torch.tensor([[1e5, -3e7, 12345.678]], device=DEV)          # beyond the fast-reduction limit: libm path
    close(f10(big), R.posenc(big.cpu(), 10), 2e-6)


@pytest.mark.parametrize("tag,D,W", [("d8w256", 8, 256), ("d4w128", 4, 128)])
def test_mlp_embedded_F6(golden, tag, D, W):
    g = golden("F6_mlp")
    sd = synthetic.make_state_dict(int(g["seed"]), D, W)
    model = NeRF(D, W, 63, 27)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    model.to(DEV)
    x = g2d(g["x"])
    assert model(x).requires_grad                             # grad mode + trainable parameters: the training path (test_gpu_train.py)
    with torch.no_grad():
        for is_fine, net in ((False, "coarse"), (True, "fine")):
            y = model(x, is_fine=is_fine)
            ref64 = R.mlp_forward(sd, f"model_{net}.", T(g["x"]), D, 63, 27, dtype=torch.float64)
            e_ref = err(T(g[f"{tag}_{net}"]), ref64)            # the reference's own fp32 rounding noise
            e_gpu = err(y, ref64)
            print(f"{tag} {net}: |gpu-fp64| {e_gpu:.2e}   |reference-fp64| {e_ref:.2e}")
            assert e_gpu <= max(4 * e_ref, 2e-5)
            close(y, g[f"{tag}_{net}"], 1e-4, 1e-5)
        # odd sizes: tail tile, a single row, nothing
        assert torch.equal(model(x[:37]), model(x)[:37])
        assert model(x[:1]).shape == (1, 4) and model(x[:0]).shape == (0, 4)


def test_mlp_repack_on_weight_change():
    sd = synthetic.make_state_dict(3, 4, 128)
    model = NeRF(4, 128, 63, 27)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    model.to(DEV)
    x = torch.rand(64, 90, device=DEV)
    with torch.no_grad():
        y0 = model(x)
        model.model_coarse.linear_color.bias.add_(1.0)
        y1 = model(x)
    close(y1[:, :3] - y0[:, :3], np.ones((64, 3), np.float32), 1e-5)


def test_submodules_are_callable_like_the_reference():
    """model.model_coarse(x) / model.model_fine(x) (model/NeRF.py:33-52: the reference's NeRF.forward only dispatches to them)."""
    sd = synthetic.make_state_dict(3, 4, 128)
    model = NeRF(4, 128, 63, 27)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    model.to(DEV)
    x = torch.rand(70, 90, device=DEV)
    with torch.no_grad():
        assert torch.equal(model.model_coarse(x), model(x)) and torch.equal(model.model_fine(x), model(x, is_fine=True))
    assert "_parent" not in model.state_dict() and len(list(model.model_coarse.children())) == 5      # no new state, no cycle of sub-modules
    y = model.model_fine(x)                                                                            # grad mode: differentiable like model(x, True)
    y.sum().backward()
    assert model.model_fine.linear_color.weight.grad is not None and model.model_coarse.linear_color.weight.grad is None


def test_weights_written_through_data_are_seen(lego_rays):
    """Writes through ``p.data`` (EMA, hand-written optimisers, weight loaders) bump no version counter: the inference path must
    still render with the new values -- packed_for() re-packs nn.Module models from the live parameters on every call."""
    sd = synthetic.make_state_dict(3, 4, 128)
    model = NeRF(4, 128, 63, 27)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    model.to(DEV)
    opts = make_opts(N_samples_c=16, N_samples_f=16)
    rays = lego_rays[:32].contiguous()
    with torch.no_grad():
        a = NP.render_rays(rays, model, None, opts, seed=3)
        v0 = model.model_fine.linear_color.bias._version
        model.model_fine.linear_color.bias.data.mul_(0.0).add_(5.0)            # sigmoid(5 + ...) ~ 1: a visibly different image
        model.model_coarse.linear_x[0].weight.data.copy_(torch.zeros_like(model.model_coarse.linear_x[0].weight))
        assert model.model_fine.linear_color.bias._version == v0               # the old cache key could not see this
        b = NP.render_rays(rays, model, None, opts, seed=3)
        fresh = weights.PackedNeRF.from_state_dict({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, DEV)
        c = NP.render_rays(rays, fresh, None, opts, seed=3)
    assert not torch.equal(a["rgb_f"], b["rgb_f"]) and not torch.equal(a["rgb_c"], b["rgb_c"])
    assert torch.equal(b["rgb_f"], c["rgb_f"]) and torch.equal(b["rgb_c"], c["rgb_c"])


@pytest.mark.parametrize("S", [64, 192, 40])
def test_mlp_rays_fused_vs_oracle(packed_big, lego_rays, S):
    sd = synthetic.make_state_dict(0, 8, 256)
    n = 50
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(2, 0, 0, n, S)) * 4 + 2, -1)[0]
    raw = ops.mlp_rays(packed_big.net, packed_big.fine, rays, z.to(DEV))
    emb = R.embed(rays.cpu(), z, 10, 4)
    ref64 = R.mlp_forward(sd, "model_fine.", emb.double(), 8, 63, 27, dtype=torch.float64).reshape(n, S, 4)
    ref32 = R.mlp_forward(sd, "model_fine.", emb, 8, 63, 27).reshape(n, S, 4)
    e_gpu, e_ref = err(raw, ref64), err(ref32, ref64)
    print(f"S={S}: |gpu-fp64| {e_gpu:.2e}  |torch fp32-fp64| {e_ref:.2e}")
    assert e_gpu <= max(4 * e_ref, 5e-5)
    # and the unfused route (embed kernel -> embedded-input kernel) agrees with the fused one
    raw2 = ops.mlp_embedded(packed_big.net, packed_big.fine, ops.embed(rays, z.to(DEV), 10, 4)).reshape(n, S, 4)
    close(raw2, raw, 2e-4, 1e-4)


@pytest.mark.parametrize("n,S", [(1500, 70), (4096, 33), (1024, 64), (1023, 64)])
def test_mlp_rays_walks_agree(packed_big, lego_rays, n, S):
    """Ray-major walk (n >= 4 x grid: whole rays per wave, the view-direction term kept across a ray's chunks), tile-major walk
    (smaller n) and the unfused route must give the same numbers: every ray is evaluated alone as a 1-ray batch (always
    tile-major) and compared bit for bit with its rows in the big launch; ragged n, odd tiles per ray, partial last tiles."""
    rays = lego_rays[:n].contiguous()
    z = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(n + S)) * 4 + 2, -1)[0].to(DEV)
    raw = ops.mlp_rays(packed_big.net, packed_big.fine, rays, z)
    assert torch.isfinite(raw).all()
    for i in (0, 1, n // 3, n // 2 + 1, n - 2, n - 1):
        one = ops.mlp_rays(packed_big.net, packed_big.fine, rays[i:i + 1].contiguous(), z[i:i + 1].contiguous())
        assert torch.equal(one[0], raw[i]), i
    # the training forward walks the same way and stores the same raw
    raw_t, _ = ops.mlp_rays_train(packed_big.net, packed_big.fine, rays, z)
    assert torch.equal(raw_t, raw)


@pytest.mark.parametrize("W,n,S", [(512, 2048, 40), (512, 1500, 40), (384, 2048, 17), (512, 1024, 16), (384, 1023, 33)])
def test_wide_kernel_walks_agree(lego_rays, W, n, S):
    """The same for mlp_fp32_wide_kernel (netWidth 384 / 512: 16-sample wave tiles): ray-major walk (2048 rays x 3 tiles on 1024 waves), tile-major
    (1500 rays: dealing whole rays would cost a round more), one tile per ray, partial last tiles -- single rays bit for bit against the big launch,
    the whole launch against the unfused route (embed + embedded-mode kernel) and, on a sample of rays, the oracle."""
    sd = synthetic.make_state_dict(60 + W, 3, W, skips=(0,))
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    rays = lego_rays[:n].contiguous()
    z = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(n + S)) * 4 + 2, -1)[0].to(DEV)
    raw = ops.mlp_rays(packed.net, packed.fine, rays, z)
    assert torch.isfinite(raw).all()
    for i in (0, 1, n // 3, n // 2 + 1, n - 2, n - 1):
        one = ops.mlp_rays(packed.net, packed.fine, rays[i:i + 1].contiguous(), z[i:i + 1].contiguous())
        assert torch.equal(one[0], raw[i]), i
    raw2 = ops.mlp_embedded(packed.net, packed.fine, ops.embed(rays, z, 10, 4)).reshape(n, S, 4)
    close(raw2, raw, 2e-4, 1e-4)
    pick = torch.tensor([0, 7, n // 2, n - 1])
    ref = R.mlp_forward(sd, "model_fine.", R.embed(rays[pick.to(DEV)].cpu(), z[pick.to(DEV)].cpu(), 10, 4).double(), 3, 63, 27, skips=(0,),
                        dtype=torch.float64).reshape(len(pick), S, 4)
    assert err(raw[pick.to(DEV)], ref) <= 2e-4
    assert ops.mlp_rays(packed.net, packed.fine, rays[:0].contiguous(), z[:0].contiguous()).shape == (0, S, 4)      # an empty batch launches nothing


@pytest.mark.parametrize("S", [64, 192])
def test_composite_F7(golden, S):
    g = golden("F7_post_process")
    out = NP.post_process(g2d(g[f"S{S}_raw"]), g2d(g[f"S{S}_z"]), g2d(g[f"S{S}_rays_d"]))
    for got, key, tol in zip(out, ("rgb", "disp", "acc", "weights", "depth"), (2e-6, 2e-6, 2e-6, 1e-6, 1e-5)):
        close(got, g[f"S{S}_{key}"], tol, 2e-6, what=f"S{S} {key}")
    assert float(out[1][0]) == 0.0 and float(out[2][0]) == 0.0 and torch.all(out[0][0] == 1.0)   # empty ray


def test_composite_kat_and_odd_sizes(golden):
    g = golden("F7_post_process")
    raw = torch.tensor([[[0, 0, 0, 1], [1, -1, 2, .5], [.5, .5, .5, -3], [2, 2, 2, 10]]], dtype=torch.float32, device=DEV)
    rgb, disp, acc, w, dep = NP.post_process(raw, torch.tensor([[2, 3, 4.5, 6]], device=DEV), torch.tensor([[0, 0, -2.]], device=DEV))
    close(rgb, g["kat_rgb"], 2e-7); close(w, g["kat_weights"], 2e-7); close(disp, g["kat_disp"], 2e-7)
    close(dep, g["kat_depth"], 5e-7); close(acc, g["kat_acc"], 2e-7)
    rs = np.random.RandomState(5)
    for S in (1, 2, 63, 65, 100, 129, 300):
        n = 9
        raw = T(rs.normal(0, 2, size=(n, S, 4)).astype(np.float32)); raw[..., 3] *= 5
        z = torch.sort(T(rs.uniform(2, 6, size=(n, S)).astype(np.float32)), -1)[0]
        d = T(rs.normal(size=(n, 3)).astype(np.float32))
        got = NP.post_process(raw.to(DEV), z.to(DEV), d.to(DEV))
        for i, (a, b) in enumerate(zip(got, R.post_process(raw, z, d))):
            if S == 1 and i == 3:       # the reference's weights tensor is EMPTY for S=1 (dists[..., :1] of an empty slice)
                assert b.shape == (n, 0) and float(a.abs().max()) == 0.0
                continue
            close(a, b, 3e-6, 3e-6, what=f"S={S}")


@pytest.mark.parametrize("tag", ["legoA", "legoA_det", "plumbP", "fernN"])
def test_render_rays_F8(golden, tag):
    """Stage-wise against the tensors captured from the reference (each stage fed the REFERENCE's inputs, so a
    stage's tolerance is its own rounding, not amplified upstream noise), then end to end.

    Why stage-wise: the path is ill-conditioned in the sample depths.  One ulp of z (5e-7; the reference's own
    torch.linspace differs by that much between its CPU-vectorised and CUDA kernels) moves the top positional
    band's phase by 2^9 * ulp and raw outputs by up to ~5e-4, and sample_pdf is discontinuous."""
    g = golden("F8_render_rays")
    D, W, Nf = int(g[f"{tag}_D"]), int(g[f"{tag}_W"]), int(g[f"{tag}_Nf"])
    sd = synthetic.make_state_dict(0, D, W)
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    opts = make_opts(near=float(g[f"{tag}_near"]), far=float(g[f"{tag}_far"]), N_samples_f=Nf, perturb=float(g[f"{tag}_perturb"]))
    posenc = (get_positional_encoder(10)[0], get_positional_encoder(4)[0])
    rays = g2d(g[f"{tag}_rays"])
    t_rand = g2d(g[f"{tag}_t_rand"])
    u = g2d(g[f"{tag}_u"]) if Nf > 0 else None
    det = opts.perturb == 0.0
    # ---- a6 stratified depths
    z_c_ref = g2d(g[f"{tag}_z_c"])
    close(ops.stratified_z(opts.near, opts.far, t_rand), z_c_ref, 1e-6, what="z_c")
    # ---- a8+a9 coarse network on the reference's depths
    raw_c = ops.mlp_rays(packed.net, packed.coarse, rays, z_c_ref)
    close(raw_c, g[f"{tag}_raw_c"], 1e-4, 1e-4, what="raw_c")
    # ---- a10 composite on the reference's raw
    raw_c_ref = g2d(g[f"{tag}_raw_c"])
    rgb_c, disp_c, _, w_c, _ = ops.composite(raw_c_ref, z_c_ref, rays)
    close(w_c, g[f"{tag}_weights_c"], 2e-6, what="weights_c")
    close(rgb_c, g[f"{tag}_rgb_c"], 2e-6, what="rgb_c staged")
    close(disp_c, g[f"{tag}_disp_c"], 2e-6, 2e-6, what="disp_c staged")
    # ---- end to end, coarse
    out = NP.render_rays(rays, packed, posenc, opts, t_rand=t_rand, u=u, return_intermediates=True)
    e_rgb_c = err(out["rgb_c"], g[f"{tag}_rgb_c"])
    print(f"{tag}: staged raw_c err {err(raw_c, g[f'{tag}_raw_c']):.2e}; end-to-end raw_c err {err(out['_raw_c'], g[f'{tag}_raw_c']):.2e}, "
          f"rgb_c err {e_rgb_c:.2e}")
    close(out["rgb_c"], g[f"{tag}_rgb_c"], 1e-4, what="rgb_c end-to-end")       # north-star bar
    close(out["disp_c"], g[f"{tag}_disp_c"], 1e-4, 1e-4, what="disp_c end-to-end")
    if Nf == 0:
        assert "rgb_f" not in out
        return
    # ---- a7 hierarchical sampling on the reference's coarse weights
    z_f_ref = g2d(g[f"{tag}_z_f"])
    z_f = ops.fine_z(z_c_ref, g2d(g[f"{tag}_weights_c"]), Nf, det, u)
    moved = float(((z_f - z_f_ref).abs() > 5e-6).float().mean())
    print(f"{tag}: staged fine depths off by >5e-6: {moved:.2e}")
    assert moved <= 5e-3, moved                          # observed 2.4e-4 .. 2.7e-3 (64-ray fixtures: one flip = 8e-5)
    # ---- fine network + composite on the reference's depths
    raw_f = ops.mlp_rays(packed.net, packed.fine, rays, z_f_ref)
    close(raw_f, g[f"{tag}_raw_f"], 1e-4, 1e-4, what="raw_f")
    rgb_f, disp_f, *_ = ops.composite(raw_f, z_f_ref, rays)
    close(rgb_f, g[f"{tag}_rgb_f"], 1e-4, what="rgb_f staged")
    close(disp_f, g[f"{tag}_disp_f"], 1e-4, 1e-4, what="disp_f staged")
    # ---- end to end, fine: identical except where sample_pdf's branches flip (the reference itself shows ~1 % of
    #      rays > 1e-4 between its fp32 and fp64 runs); report, bound, and check PSNR
    dz = (out["_z_f"].cpu() - T(g[f"{tag}_z_f"])).abs()
    ray_bad = ((out["rgb_f"].cpu() - T(g[f"{tag}_rgb_f"])).abs().max(-1)[0] > 1e-4).float().mean()
    mse = float(((out["rgb_f"].cpu() - T(g[f"{tag}_rgb_f"])) ** 2).mean())
    print(f"{tag}: end-to-end fine depths moved >1e-4: {float((dz > 1e-4).float().mean()):.2e}; rays with rgb_f off by >1e-4: "
          f"{float(ray_bad):.3f}; rgb_f max err {err(out['rgb_f'], g[f'{tag}_rgb_f']):.2e}; PSNR vs reference {R.mse2psnr(mse):.1f} dB")
    assert float((dz > 1e-4).float().mean()) <= 3e-3, float((dz > 1e-4).float().mean())     # observed <= 9.8e-4
    assert float(ray_bad) <= 0.016, float(ray_bad)                              # observed 0.000: at most ONE of the 64 rays
    assert R.mse2psnr(mse) > 100.0, R.mse2psnr(mse)                             # observed 117 .. 119 dB (the bar: PSNR within 0.05 dB)


@pytest.mark.parametrize("tag", ["blender", "llff"])
def test_batchify_F9(golden, tag):
    g = golden("F9_batchify")
    sd = synthetic.make_state_dict(3, 4, 128)
    model = NeRF(4, 128, 63, 27)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    model.to(DEV)
    opts = make_opts(near=float(g[f"{tag}_near"]), far=float(g[f"{tag}_far"]), N_samples_c=32, N_samples_f=64, chunk_rays=100, data_type=tag)
    posenc = (get_positional_encoder(10)[0], get_positional_encoder(4)[0])
    K = g[f"{tag}_K"]
    o, d = RAYS.make_o_d(16, 16, K, g2d(g[f"{tag}_pose"][:3, :4]))
    t_all, u_all = g2d(g[f"{tag}_t_rand"]), g2d(g[f"{tag}_u"])
    rc, dc, rf, df = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, 16, 16, K, opts, t_rand=t_all, u=u_all)
    assert rc.shape == (256, 3) and dc.shape == (256,) and rf.shape == (256, 3) and df.shape == (256,)
    close(rc, g[f"{tag}_rgb_c"], 1e-4, what="rgb_c"); close(dc, g[f"{tag}_disp_c"], 1e-4, 1e-4)
    bad = ((rf.cpu() - T(g[f"{tag}_rgb_f"])).abs().max(-1)[0] > 1e-4).float().mean()
    print(f"{tag}: rays with rgb_f off by >1e-4: {float(bad):.3f}")
    assert float(bad) <= 0.01, float(bad)                # observed 0.000 (256 rays)
    # K as a float64 device tensor (train.py:18) behaves the same
    rc2, *_ = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, 16, 16, torch.from_numpy(K).to(DEV), opts, t_rand=t_all, u=u_all)
    assert torch.equal(rc2, rc)
    # coarse-only returns (rgb, disp, None, None)   (nerf_process.py:252)
    opts0 = make_opts(near=opts.near, far=opts.far, N_samples_c=32, N_samples_f=0, data_type=tag)
    r = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, 16, 16, K, opts0, t_rand=t_all)
    assert r[2] is None and r[3] is None and torch.equal(r[0], rc)


def test_full_size_properties(packed_big, lego_rays):
    """BASELINE config #2 at full size (4096 rays x 64+128): properties that need no oracle run."""
    opts = make_opts()
    posenc = None
    a = NP.render_rays(lego_rays, packed_big, posenc, opts, seed=11, return_intermediates=True)
    b = NP.render_rays(lego_rays, packed_big, posenc, opts, seed=11)
    for k in ("rgb_c", "disp_c", "rgb_f", "disp_f"):
        assert torch.equal(a[k], b[k]), k                                        # deterministic
        assert torch.isfinite(a[k]).all()
    zf = a["_z_f"]
    assert torch.all(zf[:, 1:] >= zf[:, :-1]) and float(zf.min()) >= 2.0 and float(zf.max()) <= 6.0
    assert float(a["_weights_c"].sum(-1).max()) <= 1.0 + 1e-5 and float(a["_weights_c"].min()) >= 0.0
    assert float(a["rgb_f"].min()) >= -1e-6 and float(a["rgb_f"].max()) <= 1.0 + 1e-5
    assert float(a["disp_f"].min()) >= 0.0 and float(a["disp_f"].max()) <= 5.0
    # sharding invariance: the same rays rendered as 3 ragged slabs with the global ray offset -> identical bits
    parts = [NP.render_rays(lego_rays[i:j].contiguous(), packed_big, posenc, opts, seed=11, ray_offset=i)
             for i, j in ((0, 1000), (1000, 1003), (1003, 4096))]
    for k in ("rgb_c", "rgb_f", "disp_f"):
        assert torch.equal(torch.cat([p[k] for p in parts]), a[k]), k
    # a different seed changes the jitter
    c = NP.render_rays(lego_rays, packed_big, posenc, opts, seed=12)
    assert not torch.equal(c["rgb_f"], a["rgb_f"])
    # coarse samples of the full batch against the oracle on a subset (full-size launch, subset check)
    sd = synthetic.make_state_dict(0, 8, 256)
    idx = torch.arange(0, 4096, 64)
    ref = R.render_rays(lego_rays[idx].cpu(), sd, R.PathConfig(), a["_t_rand"][idx].cpu(), a["_u"][idx].cpu())
    close(a["rgb_c"][idx], ref["rgb_c"], 1e-4)
    bad = ((a["rgb_f"][idx].cpu() - ref["rgb_f"]).abs().max(-1)[0] > 1e-4).float().mean()
    assert float(bad) <= 0.016, float(bad)               # 64-ray subset; all 4096 rays: test_config2_all_rays_vs_oracle


def test_empty_and_tiny_batches(packed_big, lego_rays):
    opts = make_opts()
    out = NP.render_rays(lego_rays[:0], packed_big, None, opts, seed=1)
    assert out["rgb_f"].shape == (0, 3) and out["disp_c"].shape == (0,)
    one = NP.render_rays(lego_rays[:1].contiguous(), packed_big, None, opts, seed=1)
    many = NP.render_rays(lego_rays[:7].contiguous(), packed_big, None, opts, seed=1)
    assert torch.equal(one["rgb_f"], many["rgb_f"][:1])


def test_pre_process_surface(packed_big, lego_rays):
    opts = make_opts()
    posenc = (get_positional_encoder(10)[0], get_positional_encoder(4)[0])
    rays = lego_rays[:32].contiguous()
    t = ops.fill_uniform(4, 0, 0, 32, 64, DEV)
    emb, z, rd = NP.pre_process(rays, posenc, opts, isFine=False, t_rand=t)
    assert emb.shape == (32 * 64, 90) and z.shape == (32, 64) and torch.equal(rd, rays[:, 3:])
    raw = NP.run_network(packed_big, emb).reshape(32, 64, 4)
    rgb, disp, acc, w, depth = NP.post_process(raw, z, rd)
    u = ops.fill_uniform(4, 1, 0, 32, 128, DEV)
    emb_f, z_f, _ = NP.pre_process(rays, posenc, opts, z_vals=z, weights=w, isFine=True, u=u)
    assert emb_f.shape == (32 * 192, 90) and z_f.shape == (32, 192)
    # the staged surface and the fused entry agree
    fused = NP.render_rays(rays, packed_big, posenc, opts, t_rand=t, u=u, return_intermediates=True)
    close(fused["_z_c"], z, 0); close(fused["rgb_c"], rgb, 2e-4)
    assert float((fused["_z_f"] - z_f).abs().max()) <= 1e-3


# ---------------------------------------------------------------------------------------------------
# bf16 MFMA variant (BASELINE config #5): bf16 weights/activations, fp32 accumulate -- compared with the fp32 path
# ---------------------------------------------------------------------------------------------------
def test_bf16_mlp_vs_fp32(packed_big, lego_rays):
    n, S = 64, 192
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(2, 0, 0, n, S)) * 4 + 2, -1)[0].to(DEV)
    raw32 = ops.mlp_rays(packed_big.net, packed_big.fine, rays, z)
    raw16 = ops.mlp_rays(packed_big.net, packed_big.bf16()[1], rays, z, bf16=True)
    d = (raw16 - raw32).abs()
    scale = float(raw32.abs().mean())
    rel = float(d.mean()) / scale
    print(f"bf16 vs fp32 raw: mean |diff| {float(d.mean()):.3e} (mean |raw| {scale:.3f}, relative {rel:.2e}), max {float(d.max()):.3e}")
    assert torch.isfinite(raw16).all()
    assert rel < 1e-2 and float(d.max()) < 0.2, (rel, float(d.max()))      # observed 6.5e-3 / 6.6e-2: ~8 mantissa bits per product over 256-term sums


def test_bf16_render_psnr(packed_big, lego_rays):
    opts = make_opts()
    a = NP.render_rays(lego_rays, packed_big, None, opts, seed=3, return_intermediates=True)
    b = NP.render_rays(lego_rays, packed_big, None, opts, t_rand=a["_t_rand"], u=a["_u"], bf16=True)
    for k in ("rgb_c", "rgb_f"):
        mse = float(((a[k] - b[k]) ** 2).mean())
        print(f"bf16 vs fp32 {k}: PSNR {R.mse2psnr(mse):.1f} dB, max |diff| {float((a[k] - b[k]).abs().max()):.3e}")
        assert torch.isfinite(b[k]).all() and R.mse2psnr(mse) > (70.0 if k == "rgb_c" else 50.0), (k, R.mse2psnr(mse))      # observed 74.6 / 53.9 dB
    # odd sizes and determinism
    c = NP.render_rays(lego_rays[:7].contiguous(), packed_big, None, opts, t_rand=a["_t_rand"][:7], u=a["_u"][:7], bf16=True)
    assert torch.equal(c["rgb_f"], b["rgb_f"][:7])


# ---------------------------------------------------------------------------------------------------
# other network shapes, the frame harness, error behaviour
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,W", [(6, 128), (2, 128), (3, 256), (10, 256)])
def test_other_network_shapes(D, W):
    """skip=[4] only fires when D >= 6 (model/NeRF.py:25 builds range(D-1)); depth is a run-time loop in the kernel."""
    sd = synthetic.make_state_dict(21, D, W)
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    assert packed.net.skip == (4 if D >= 6 else -1)
    x = torch.rand(100, 90) * 2 - 1
    y = ops.mlp_embedded(packed.net, packed.coarse, x.to(DEV))
    ref = R.mlp_forward(sd, "model_coarse.", x, D, 63, 27, dtype=torch.float64)
    assert err(y, ref) <= 5e-5
    rays = torch.cat([torch.rand(9, 3) * 2 - 1, torch.nn.functional.normalize(torch.randn(9, 3), dim=-1)], -1)
    z = torch.sort(torch.rand(9, 64) * 4 + 2, -1)[0]
    raw = ops.mlp_rays(packed.net, packed.fine, rays.to(DEV), z.to(DEV))
    ref = R.mlp_forward(sd, "model_fine.", R.embed(rays, z, 10, 4).double(), D, 63, 27, dtype=torch.float64).reshape(9, 64, 4)
    assert err(raw, ref) <= 1e-4


@pytest.mark.parametrize("D,W,skip", [(8, 64, 4), (4, 64, 4), (8, 192, 4), (6, 100, 2), (8, 200, 4), (3, 31, 0), (8, 129, 4),
                                      (8, 512, 4), (4, 300, 1), (8, 257, 4), (2, 512, -1), (16, 384, 7), (8, 400, 4), (5, 384, 2)])
def test_network_widths_without_a_kernel_of_their_own(D, W, skip, lego_rays):
    """--netWidth values the reference accepts (config.py:57; model/NeRF.py:24-30 builds any W) but no kernel is instantiated for: the
    packer lays such a network out for the next kernel width (64 -> 128, 192 -> 256, 300 -> 384, 400 -> 512) with zero weights for the hidden units it
    does not have (csrc/layout.h kernel_width): the kernels' results are the W-wide network's.  Wider than 256 runs on mlp_fp32_wide.hip
    (16 points per wave on v_mfma_f32_16x16x4_f32; --netWidth 384 and 512 are native widths of that kernel).  F6-style against the oracle: embedded rows, the fused
    rays entry, the whole render_rays step, a module model through batchify; bf16 and split precision run widths up to 256 (padded to their one kernel width)."""
    from nerf_pytorch_paeng_amd._lib import MiNerfError
    from nerf_pytorch_paeng_amd.model import NeRF
    skips = (skip,) if skip >= 0 else ()
    sd = synthetic.make_state_dict(40 + W, D, W, skips=skips)
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    assert (packed.net.D, packed.net.W) == (D, W) and packed.net.skip == (skip if 0 <= skip and skip + 1 < D else -1)
    x = torch.rand(333, 90, generator=torch.Generator().manual_seed(W)) * 2 - 1
    for fine, blob in ((False, packed.coarse), (True, packed.fine)):
        y = ops.mlp_embedded(packed.net, blob, x.to(DEV))
        ref = R.mlp_forward(sd, "model_fine." if fine else "model_coarse.", x, D, 63, 27, skips=skips, dtype=torch.float64)
        assert err(y, ref) <= 5e-5, (fine, err(y, ref))
    n, S = 70, 96
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(4, 0, 0, n, S)) * 4 + 2, -1)[0]
    raw = ops.mlp_rays(packed.net, packed.fine, rays, z.to(DEV))
    ref = R.mlp_forward(sd, "model_fine.", R.embed(rays.cpu(), z, 10, 4).double(), D, 63, 27, skips=skips, dtype=torch.float64).reshape(n, S, 4)
    assert err(raw, ref) <= 2e-4, err(raw, ref)
    # the whole step, depths pinned to the oracle's (sample_pdf is discontinuous), through a module model built at this width
    opts = make_opts()
    model = NeRF(D, W, 63, 27, skips=list(skips)).to(DEV)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    g = torch.Generator().manual_seed(3)
    t_rand, u = torch.rand(n, 64, generator=g), torch.rand(n, 128, generator=g)
    cfg = R.PathConfig(netDepth=D, netWidth=W, skips=skips)
    want = R.render_rays(rays.cpu(), sd, cfg, t_rand, u)
    with torch.no_grad():
        got = NP.render_rays(rays, model, None, opts, t_rand=t_rand, u=u)
        rc, dc, rf, df = NP.batchify_rays_and_render_by_chunk(rays[:, :3], rays[:, 3:], model, None, 8, 8, np.eye(3), opts, t_rand=t_rand, u=u)
    assert torch.equal(rc, got["rgb_c"]) and torch.equal(rf, got["rgb_f"])
    assert err(got["rgb_c"], want["rgb_c"]) <= 2e-5 and err(got["disp_c"], want["disp_c"]) <= 2e-5
    z_f = want["_z_f"].to(DEV)
    pinned = ops.composite(ops.mlp_rays(packed.net, packed.fine, rays, z_f), z_f, rays)[0]
    assert err(pinned, want["rgb_f"]) <= 2e-5
    assert float(((got["rgb_f"].cpu() - want["rgb_f"]).abs().amax(-1) > 1e-4).float().mean()) <= 0.03
    # the bf16 and split-precision variants (one kernel width, 256): narrower networks padded by weights.PackedNeRF, wider ones refused
    if W <= 256:
        for src in (packed, model):                                       # host packer (state dict) and device packer (live module)
            with torch.no_grad():
                g16 = NP.render_rays(rays, src, None, opts, t_rand=t_rand, u=u, f16s=True)
                gb = NP.render_rays(rays, src, None, opts, t_rand=t_rand, u=u, bf16=True)
            assert err(g16["rgb_c"], want["rgb_c"]) <= 2e-5, err(g16["rgb_c"], want["rgb_c"])              # fp32-grade
            assert float(((g16["rgb_f"].cpu() - want["rgb_f"]).abs().amax(-1) > 1e-4).float().mean()) <= 0.03
            mse = float(torch.mean((gb["rgb_c"].cpu() - want["rgb_c"]) ** 2))
            assert torch.isfinite(gb["rgb_f"]).all() and -10.0 * np.log10(max(mse, 1e-20)) > 50.0, mse       # bf16-grade (observed 70-80 dB coarse)
    else:
        for kw in (dict(bf16=True), dict(f16s=True)):
            with pytest.raises(MiNerfError, match="netWidth <= 256"):
                NP.render_rays(rays, packed, None, opts, t_rand=t_rand, u=u, **kw)
    if W > 256:                                                  # training: up to 256 (padded, tests/test_gpu_train.py); wider is inference only
        with pytest.raises((MiNerfError, RuntimeError), match="training kernels exist for netWidth <= 256"):
            NP.render_rays(rays, model, None, opts, t_rand=t_rand, u=u)["rgb_c"].sum().backward()


@pytest.mark.parametrize("W,L_x,L_d", [(256, 6, 2), (256, 10, 0), (256, 0, 4), (128, 4, 4), (128, 8, 1)])
def test_fewer_encoding_frequencies(W, L_x, L_d, lego_rays):
    """--L_x / --L_d below the defaults (config.py:54-55).  gamma_L is a prefix of gamma_10 in the reference's channel order
    (PositionalEncoding.py:18-24), so such a network runs on the kernels built for 10 / 4 with zero weights on the frequencies it
    does not have: embedded rows of the network's own width, the fused rays path, the bf16 variant, the training gradients."""
    D = 8 if W == 256 else 4
    in_x, in_d = 3 + 6 * L_x, 3 + 6 * L_d
    sd = synthetic.make_state_dict(41 + L_x, D, W, in_x=in_x, in_d=in_d)
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    assert (packed.net.L_x, packed.net.L_d, packed.net.skip) == (L_x, L_d, 4 if D >= 6 else -1)
    n, S = 19, 48
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(8, 0, 0, n, S)) * 4 + 2, -1)[0]
    x = R.embed(rays.cpu(), z, L_x, L_d)                                   # [n*S, in_x + in_d]: the network's own row width
    assert x.shape[1] == in_x + in_d
    ref = R.mlp_forward(sd, "model_fine.", x.double(), D, in_x, in_d, dtype=torch.float64)
    y = ops.mlp_embedded(packed.net, packed.fine, x.to(DEV))
    assert err(y, ref) <= 1e-4, err(y, ref)
    assert torch.equal(ops.embed(rays, z.to(DEV), L_x, L_d).cpu()[:, :3], x[:, :3])
    raw = ops.mlp_rays(packed.net, packed.fine, rays, z.to(DEV))
    assert err(raw.reshape(-1, 4), ref) <= 2e-4, err(raw.reshape(-1, 4), ref)
    if W == 256:
        ref16 = R.mlp_forward_bf16(sd, "model_fine.", x, D, in_x, in_d)
        raw16 = ops.mlp_rays(packed.net, packed.bf16()[1], rays, z.to(DEV), bf16=True).cpu().reshape(-1, 4)
        e = (raw16 - ref16).abs()
        rel = max(float(e[:, c].mean() / ref16[:, c].abs().mean()) for c in range(4))
        assert rel < 2e-3 and float(e.max()) < 0.2, (rel, float(e.max()))
    # the drop-in surface with a model of that shape, inference and one training step's gradients
    model = NeRF(D, W, in_x, in_d).to(DEV)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    opts = make_opts(N_samples_c=16, N_samples_f=16)
    posenc = (get_positional_encoder(L_x)[0], get_positional_encoder(L_d)[0])
    with torch.no_grad():
        out = NP.render_rays(rays, model, posenc, opts, seed=2, return_intermediates=True)
    want = R.render_rays(rays.cpu(), sd, R.PathConfig(N_samples_c=16, N_samples_f=16, L_x=L_x, L_d=L_d, netDepth=D, netWidth=W),
                         out["_t_rand"].cpu(), out["_u"].cpu())
    assert err(out["rgb_c"], want["rgb_c"]) <= 2e-5
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(1))
    psd = {k: T(v).clone().float().requires_grad_(True) for k, v in sd.items()}
    ref_t = R.render_rays(rays.cpu(), psd, R.PathConfig(N_samples_c=16, N_samples_f=16, L_x=L_x, L_d=L_d, netDepth=D, netWidth=W),
                          out["_t_rand"].cpu(), out["_u"].cpu())
    torch.mean((ref_t["rgb_c"] - tgt) ** 2).backward()
    got = NP.render_rays(rays, model, posenc, opts, t_rand=out["_t_rand"], u=out["_u"])
    torch.mean((got["rgb_c"] - tgt.to(DEV)) ** 2).backward()
    for k, p_ in model.model_coarse.named_parameters():
        g_ref = psd["model_coarse." + k].grad
        assert p_.grad.shape == g_ref.shape
        assert float((p_.grad.cpu() - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max()) + 1e-9, k
    with pytest.raises(Exception):
        weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(1, 4, 128, in_x=3 + 6 * 11), DEV)      # more than 10 frequencies


def test_frame_harness_rows_and_llff(packed_big):
    """dist.render_frame (the counterpart of test.py:38-53) on one process: a frame rendered whole equals the same frame
    rendered as two row blocks with global ray offsets; llff mode runs the NDC warp."""
    from nerf_pytorch_paeng_amd import dist as mdist
    for data_type, (K, H, W), pose, nf in (("blender", synthetic.lego_camera(), synthetic.pose_spherical(30.0, -30.0, 4.0), (2.0, 6.0)),
                                             ("llff", synthetic.fern_camera(), synthetic.fern_pose(), (0.0, 1.0))):
        s = 24.0 / W
        Ks = K.copy(); Ks[0, 0] *= s; Ks[1, 1] *= s; Ks[0, 2] = 12.0; Ks[1, 2] = 10.0
        h, w = 20, 24
        opts = make_opts(near=nf[0], far=nf[1], data_type=data_type)
        rgb, disp = mdist.render_frame(h, w, Ks, pose, packed_big, opts, seed=5)
        assert rgb.shape == (h, w, 3) and disp.shape == (h, w) and torch.isfinite(rgb).all()
        top = mdist.render_frame(h, w, Ks, pose, packed_big, opts, seed=5, render_rows_fn=None)[0]
        assert torch.equal(top, rgb)
        # two row blocks by hand
        parts = []
        for r0, nr in ((0, 7), (7, 13)):
            _, d = ops.make_o_d(w, h, Ks, pose, DEV, row0=r0, n_rows=nr, want_origins=False)
            o = torch.as_tensor(pose)[:3, -1].to(DEV).expand(d.shape)
            out = NP.batchify_rays_and_render_by_chunk(o, d, packed_big, None, h, w, Ks, opts, seed=5, ray_offset=r0 * w)
            parts.append(out[2])
        assert torch.equal(torch.cat(parts, 0).reshape(h, w, 3), rgb)


def test_error_behaviour(packed_big, lego_rays):
    from nerf_pytorch_paeng_amd._lib import MiNerfError
    opts = make_opts()
    with pytest.raises(MiNerfError):
        NP.render_rays(lego_rays[:4, :5].contiguous(), packed_big, None, opts)                 # not [n, 6]
    with pytest.raises(MiNerfError):
        ops.mlp_embedded(packed_big.net, packed_big.coarse, torch.rand(4, 63, device=DEV))        # wrong channel count
    with pytest.raises(MiNerfError):
        ops.sample_pdf(torch.rand(4, 8, device=DEV), torch.rand(4, 8, device=DEV), 16, False, None)   # weights must be B-1, u needed
    with pytest.raises(MiNerfError):
        weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 4, 600), DEV)              # wider than the widest kernel (narrower ones pad)
    with pytest.raises(MiNerfError):
        NP.render_rays(lego_rays[:4].cpu(), weights.packed_for(packed_big), None, opts, t_rand=torch.rand(4, 63))   # wrong t_rand shape


def test_bf16_through_module_model(lego_rays):
    """nn.Module models are re-packed on the device; their bf16 blobs are packed lazily from the module's state dict."""
    sd = synthetic.make_state_dict(0, 8, 256)
    model = NeRF(8, 256, 63, 27).to(DEV)
    model.load_state_dict({k: T(v) for k, v in sd.items()})
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    opts = make_opts()
    rays = lego_rays[:64].contiguous()
    with torch.no_grad():
        a = NP.render_rays(rays, model, None, opts, seed=3, bf16=True)
        b = NP.render_rays(rays, packed, None, opts, seed=3, bf16=True)
        c = NP.render_rays(rays, model, None, opts, seed=3)
        d = NP.render_rays(rays, packed, None, opts, seed=3)
    assert torch.equal(a["rgb_f"], b["rgb_f"]) and torch.equal(c["rgb_f"], d["rgb_f"])


def test_bf16_mlp_vs_bf16_oracle(packed_big, lego_rays):
    """The bf16 kernel against the oracle with the kernel's rounding points (R.mlp_forward_bf16: bf16 weights, bf16 gamma(x), bf16
    activations, fp32 accumulation) -- the check that a systematic packing / ordering error cannot pass: such an error moves
    outputs by O(1e-1) of their scale, fp32 summation order and round-to-bf16 boundary flips move them by O(1e-3).  Both launch
    shapes (64 and 32 points per wave; the second serves small launches such as a 512-ray shard) and the launcher's own choice."""
    sd = synthetic.make_state_dict(0, 8, 256)
    for n, S in ((64, 192), (33, 100), (5, 64), (512, 64), (512, 192)):
        rays = lego_rays[:n].contiguous()
        z = torch.sort(T(R.counter_uniform(2, 0, 0, n, S)) * 4 + 2, -1)[0]
        ref = R.mlp_forward_bf16(sd, "model_fine.", R.embed(rays.cpu(), z, 10, 4), 8, 63, 27).reshape(n, S, 4)
        outs = {}
        for ppw in (64, 32, 0):
            raw16 = ops.mlp_rays(packed_big.net, packed_big.bf16()[1], rays, z.to(DEV), bf16=True, points_per_wave=ppw).cpu()
            outs[ppw] = raw16
            e = (raw16 - ref).abs()
            rel = [float(e[..., c].mean() / ref[..., c].abs().mean()) for c in range(4)]
            print(f"bf16 kernel ({ppw or 'auto'} points per wave) vs bf16 oracle n={n} S={S}: mean |err| / mean |ref| per channel "
                  f"{[f'{v:.1e}' for v in rel]}, max |err| {float(e.max()):.3e}")
            assert torch.isfinite(raw16).all()
            assert max(rel) < 2e-3, rel                      # observed 2e-4 .. 8e-4
            assert float(e.max()) < 0.2, float(e.max())      # a flipped bf16 rounding of one activation, amplified by the x20 density head
        # a point's arithmetic does not depend on the shape of the launch that carries it
        assert torch.equal(outs[64], outs[32]) and torch.equal(outs[0], outs[64])


@pytest.mark.parametrize("n,S", [(341, 192), (342, 192), (700, 192), (1366, 192), (1500, 192), (2731, 192), (4096, 192), (1000, 100), (3000, 33), (171, 192)])
def test_bf16_launch_plans_agree(n, S, packed_big, lego_rays):
    """The launcher's plan (whole rounds of the 64-point shape + the remainder as one round of the 32-point shape, split at a TILE, not
    a ray, boundary; ray-major or flat tile walk) against each shape pinned over the whole call: every point of every ray written,
    bit-identical.  Sizes straddle the round boundaries of a 256-CU chip (341.33 rays x 192 samples = one 64-point round)."""
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(6, 0, 0, n, S)) * 4 + 2, -1)[0].to(DEV)
    outs = []
    for ppw in (64, 32, 0):
        raw = torch.full((n, S, 4), float("nan"), device=DEV)
        args = (__import__("ctypes").byref(packed_big.net), packed_big.bf16()[1].data_ptr(), rays.data_ptr(), z.data_ptr(), n, S, raw.data_ptr(), ppw,
                torch.cuda.current_stream(DEV).cuda_stream)
        from nerf_pytorch_paeng_amd._lib import check, lib
        check(lib().mi_nerf_mlp_rays_bf16_shape(*args), "mi_nerf_mlp_rays_bf16_shape")
        assert torch.isfinite(raw).all(), (n, S, ppw)          # no point left unwritten by a split launch
        outs.append(raw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_bf16_small_launch_shape_whole_step(packed_big, lego_rays):
    """A 512-ray shard (what one of 8 GPUs renders of BASELINE config #5's batch) through render_rays with the launch shape chosen
    by the launcher, pinned to 64 and pinned to 32 points per wave: identical outputs; PSNR against the fp32 path as for 4096 rays."""
    opts = make_opts()
    rays = lego_rays[:512].contiguous()
    a = NP.render_rays(rays, packed_big, None, opts, seed=3, return_intermediates=True)
    blobs = packed_big.bf16()
    outs = {}
    for ppw in (0, 64, 32):
        cfg = ops.render_cfg(opts.near, opts.far, 64, 128, False, True, points_per_wave=ppw)
        outs[ppw] = ops.render_rays(packed_big.net, blobs[0], blobs[1], cfg, rays, a["_t_rand"], a["_u"])[:4]
    for ppw in (64, 32):
        for x, y in zip(outs[0], outs[ppw]):
            assert torch.equal(x, y), ppw
    for k, got in (("rgb_c", outs[0][0]), ("rgb_f", outs[0][2])):
        mse = float(((a[k] - got) ** 2).mean())
        print(f"bf16 (512-ray shard) vs fp32 {k}: PSNR {R.mse2psnr(mse):.1f} dB")
        assert torch.isfinite(got).all() and R.mse2psnr(mse) > (70.0 if k == "rgb_c" else 50.0), (k, R.mse2psnr(mse))
    assert torch.equal(NP.render_rays(rays, packed_big, None, opts, t_rand=a["_t_rand"], u=a["_u"], bf16=True)["rgb_f"], outs[0][2])


@pytest.mark.parametrize("n,Sc,Nf,det,inject", [(512, 64, 128, False, False), (511, 64, 128, False, True), (1, 64, 128, False, False), (2, 64, 128, True, False),
                                                (333, 40, 17, False, True), (300, 64, 192, False, False), (512, 33, 64, True, False), (513, 64, 128, False, False),
                                                (700, 64, 128, False, True), (256, 64, 0, False, False), (100, 32, 32, False, False),
                                                (342, 64, 128, False, False), (343, 64, 128, False, True), (400, 64, 100, True, False), (450, 64, 97, False, False),
                                                (341, 64, 128, False, False), (512, 64, 130, False, False), (1024, 64, 128, False, False), (1023, 64, 128, False, True),
                                                (1025, 64, 128, False, False), (1000, 48, 64, True, False)])
def test_bf16_small_coarse_launch_does_the_middle_of_render_rays_itself(n, Sc, Nf, det, inject, packed_big, lego_rays):
    """Round 6: a small bf16 coarse launch (one 32-point unit per wave: up to 512 rays on 256 CUs, 33..64 coarse samples) composites the two rays each
    workgroup owns and draws their fine depths in the kernel's epilogue -- the device functions of composite_fine_z_kernel, called in place -- so the
    step has one launch fewer; from 513 to 1024 rays (one 64-point unit per wave = one unit per ray) every wave does that for the ray it computed.  (The same for the LAST compositing -- the fine launch dealing its tiles so that a workgroup owns two rays -- was built
    and measured in round 6 and is not shipped: +1 us at 512 rays, profiles/r06_bf16_fused_stages_ab.txt.)  Every output and intermediate equals, bit for bit, the same step with the launch shape pinned (64 or 32 points per
    wave: the pinned forms never fuse and run the stage kernel): odd ray counts (a workgroup with one ray), one ray, ragged sample counts,
    deterministic and injected jitter, both ownership forms (<= 512 rays: two rays per workgroup; 513..1024: a ray per wave), sizes and shapes that must NOT
    fuse (1025 rays, 32 coarse samples = one tile per ray, coarse only)."""
    opts = make_opts(N_samples_c=Sc, N_samples_f=Nf, perturb=0.0 if det else 1.0)
    rays = lego_rays[:n].contiguous()
    blobs = packed_big.bf16()
    t_rand = ops.fill_uniform(5, 0, 7, n, Sc, DEV) if inject else None
    u = ops.fill_uniform(5, 1, 7, n, Nf, DEV) if (inject and Nf > 0 and not det) else None
    res = {}
    for ppw in (0, 32, 64):
        cfg = ops.render_cfg(opts.near, opts.far, Sc, Nf, det, True, points_per_wave=ppw, seed=5, ray_offset=7)
        ws = torch.full((ops.workspace_layout(cfg, n).total,), 0xFF, dtype=torch.uint8, device=DEV)      # NaN bit patterns: an unwritten slot shows
        out = ops.render_rays(packed_big.net, blobs[0], blobs[1] if Nf > 0 else None, cfg, rays, t_rand, u, workspace=ws)
        res[ppw] = (out[:4], {k: v.clone() for k, v in ops.workspace_views(cfg, n, out[4]).items()})
    for ppw in (32, 64):
        for x, y in zip(res[0][0], res[ppw][0]):
            assert (x is None and y is None) or torch.equal(x, y), (ppw, n, Sc, Nf)
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[ppw][1][k]), (k, ppw, n, Sc, Nf)
    assert all(torch.isfinite(v).all() for v in res[0][1].values())
    if Nf > 0:
        z_f = res[0][1]["z_f"]
        assert bool((z_f[:, 1:] >= z_f[:, :-1]).all()) and torch.isfinite(res[0][0][2]).all()


@pytest.mark.parametrize("D,skip", [(7, 5), (3, -1), (8, 3), (2, -1), (6, 4), (9, 0)])
def test_bf16_other_depths_and_skip_positions(D, skip, lego_rays):
    """The bf16 kernel addresses its B fragments by explicit AGPR number, with two statically unrolled polarities (which fragment
    set a layer reads) and a separate instantiation for the skip layer at odd / even l -- a polarity error would give silently
    wrong colours.  check_net_bf16 accepts D = 2..16 and any skip position (config.py:54-57 --netDepth; NeRF.py:25 skips), so each
    code path runs here against the bf16 oracle: odd D (tail reads set 0), even D (set 1), the skip layer at even l
    (skip = 3, 5: layer_10<SKIP>) and at odd l (skip = 0, 4: layer_01<SKIP>), no skip at all."""
    skips = (skip,) if skip >= 0 else ()
    sd = synthetic.make_state_dict(31 + D, D, 256, skips=skips)
    packed = weights.PackedNeRF.from_state_dict(sd, DEV)
    assert packed.net.D == D and packed.net.skip == skip
    n, S = 37, 96
    rays = lego_rays[:n].contiguous()
    z = torch.sort(T(R.counter_uniform(4, 0, 0, n, S)) * 4 + 2, -1)[0]
    raw16 = ops.mlp_rays(packed.net, packed.bf16()[1], rays, z.to(DEV), bf16=True, points_per_wave=64).cpu()
    assert torch.equal(raw16, ops.mlp_rays(packed.net, packed.bf16()[1], rays, z.to(DEV), bf16=True, points_per_wave=32).cpu())
    ref = R.mlp_forward_bf16(sd, "model_fine.", R.embed(rays.cpu(), z, 10, 4), D, 63, 27, skips=skips).reshape(n, S, 4)
    e = (raw16 - ref).abs()
    rel = [float(e[..., c].mean() / ref[..., c].abs().mean()) for c in range(4)]
    print(f"bf16 kernel vs bf16 oracle D={D} skip={skip}: mean |err| / mean |ref| per channel {[f'{v:.1e}' for v in rel]}, max |err| {float(e.max()):.3e}")
    assert torch.isfinite(raw16).all()
    assert max(rel) < 2e-3 and float(e.max()) < 0.2, (rel, float(e.max()))
    # and the fp32 kernel on the same network (its depth / skip position are run-time values)
    raw32 = ops.mlp_rays(packed.net, packed.fine, rays, z.to(DEV)).cpu()
    ref32 = R.mlp_forward(sd, "model_fine.", R.embed(rays.cpu(), z, 10, 4).double(), D, 63, 27, skips=skips, dtype=torch.float64).reshape(n, S, 4)
    assert err(raw32, ref32) <= 2e-4, err(raw32, ref32)


@pytest.mark.parametrize("bf16", [False, True])
def test_in_kernel_jitter_equals_explicit_tensors(bf16, packed_big, lego_rays):
    """render_rays draws its jitter inside the sampling kernels when no t_rand / u tensor is passed (the reference draws torch.rand
    inside pre_process / sample_pdf, nerf_process.py:58-60,162-163): the same counter-based values mi_nerf_fill_uniform writes, keyed
    on (seed, ray_offset + ray, sample) -- so outputs are bit-identical to the explicit-tensor call, for any shard of the batch.
    Also: the fused composite + resample + merge launch of the fused entry equals the staged kernels bit for bit."""
    opts = make_opts()
    n, first = 700, 1234
    rays = lego_rays[first:first + n].contiguous()
    a = NP.render_rays(rays, packed_big, None, opts, seed=9, ray_offset=first, bf16=bf16)
    t_rand, u = ops.fill_uniform(9, 0, first, n, 64, DEV), ops.fill_uniform(9, 1, first, n, 128, DEV)
    b = NP.render_rays(rays, packed_big, None, opts, t_rand=t_rand, u=u, bf16=bf16, return_intermediates=True)
    c = NP.render_rays(rays, packed_big, None, opts, seed=9, ray_offset=first, bf16=bf16, return_intermediates=True)
    for k in ("rgb_c", "disp_c", "rgb_f", "disp_f"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    assert torch.equal(c["_t_rand"], t_rand) and torch.equal(c["_u"], u)
    # staged: composite and fine_z as their own launches on the fused call's coarse outputs
    rgb_c, disp_c, _, w_c, _ = ops.composite(b["_raw_c"], b["_z_c"], rays)
    assert torch.equal(rgb_c, b["rgb_c"]) and torch.equal(disp_c, b["disp_c"]) and torch.equal(w_c, b["_weights_c"])
    assert torch.equal(ops.fine_z(b["_z_c"], w_c, 128, False, u), b["_z_f"])
    assert torch.equal(ops.stratified_z(opts.near, opts.far, t_rand), b["_z_c"])
    # deterministic sampling ignores u; coarse-only skips the resampling
    d0 = NP.render_rays(rays, packed_big, None, make_opts(perturb=0.0), seed=9, ray_offset=first, bf16=bf16, return_intermediates=True)
    assert d0["_u"] is None and torch.equal(ops.fine_z(d0["_z_c"], d0["_weights_c"], 128, True, None), d0["_z_f"])
    e0 = NP.render_rays(rays, packed_big, None, make_opts(N_samples_f=0), seed=9, ray_offset=first, bf16=bf16)
    assert "rgb_f" not in e0 and torch.equal(e0["rgb_c"], a["rgb_c"])
    # sample counts that are not multiples of the 32-sample tile (the bf16 kernel draws the coarse depths in its own prologue)
    for n2, Sc, Nf in ((77, 40, 24), (3, 33, 7), (130, 16, 16)):
        o2 = make_opts(N_samples_c=Sc, N_samples_f=Nf)
        r2 = lego_rays[:n2].contiguous()
        x = NP.render_rays(r2, packed_big, None, o2, seed=4, ray_offset=99, bf16=bf16, return_intermediates=True)
        y = NP.render_rays(r2, packed_big, None, o2, seed=4, ray_offset=99, bf16=bf16)
        assert torch.equal(x["_z_c"], ops.stratified_z(o2.near, o2.far, ops.fill_uniform(4, 0, 99, n2, Sc, DEV)))
        assert torch.equal(x["rgb_f"], y["rgb_f"]) and torch.equal(x["disp_c"], y["disp_c"]) and torch.isfinite(y["rgb_f"]).all()


def test_config2_all_rays_vs_oracle(packed_big, lego_rays, oracle_cache):
    """BASELINE config #2 at full size, EVERY ray against the CPU oracle: coarse colours and disparities directly; fine outputs
    with the depths pinned to the ones the HIP path sampled (sample_pdf's branch flips are counted separately below)."""
    opts = make_opts()
    a = NP.render_rays(lego_rays, packed_big, None, opts, seed=11, return_intermediates=True)
    sd = synthetic.make_state_dict(0, 8, 256)
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    rc, tr, uu = lego_rays.cpu(), a["_t_rand"].cpu(), a["_u"].cpu()
    with torch.no_grad():
        if "config2_seed11" not in oracle_cache:             # the same rays and jitter as tests/test_gpu_f16s.py::test_f16s_config2_all_rays_vs_oracle
            oracle_cache["config2_seed11"] = (tr, uu, R.render_rays(rc, sd, R.PathConfig(), tr, uu))
        tr0, uu0, ref = oracle_cache["config2_seed11"]
        assert torch.equal(tr0, tr) and torch.equal(uu0, uu)
        pin = R.render_rays(rc, sd, R.PathConfig(), tr, uu, z_fine_override=a["_z_f"].cpu())
    e_c = float((a["rgb_c"].cpu() - ref["rgb_c"]).abs().max())
    e_dc = float((a["disp_c"].cpu() - ref["disp_c"]).abs().max())
    e_f = float((a["rgb_f"].cpu() - pin["rgb_f"]).abs().max())
    e_df = float(((a["disp_f"].cpu() - pin["disp_f"]).abs() / pin["disp_f"].abs().clamp_min(1e-3)).max())
    bad = float(((a["rgb_f"].cpu() - ref["rgb_f"]).abs().max(-1)[0] > 1e-4).float().mean())
    mse = float(((a["rgb_f"].cpu() - ref["rgb_f"]) ** 2).mean())
    print(f"config #2, 4096 rays vs oracle: rgb_c max err {e_c:.2e}, disp_c {e_dc:.2e}; pinned rgb_f {e_f:.2e}, disp_f rel {e_df:.2e}; "
          f"un-pinned rays off by >1e-4: {bad:.4f}, PSNR {R.mse2psnr(mse):.1f} dB")
    assert e_c <= 2e-5 and e_dc <= 2e-4, (e_c, e_dc)      # north-star bar 1e-4; observed 2e-6 / 3e-5
    assert e_f <= 2e-5 and e_df <= 2e-4, (e_f, e_df)
    assert bad <= 0.01 and R.mse2psnr(mse) > 90.0, (bad, R.mse2psnr(mse))
