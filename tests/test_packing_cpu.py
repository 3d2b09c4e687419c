"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every declared
symbol, the weight packer produces a blob whose lane-level emulation (tests/blob_emulator.py)
reproduces the oracle MLP, argument validation fails loudly.  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import _lib, ops, synthetic, weights
from oracle import restate as R
from tests.blob_emulator import Emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi_nerf.h")).read()
    declared = set(re.findall(r"\b(mi_nerf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mi_nerf_net", "mi_nerf_params"}
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(handle, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib().mi_nerf_abi_version() == _lib.ABI_VERSION


def test_infer_net_and_blob_sizes():
    for D, W, skip in ((8, 256, 4), (4, 128, -1), (6, 128, 4)):
        sd = synthetic.make_state_dict(1, D, W)
        net = weights.infer_net(sd)
        assert (net.D, net.W, net.skip, net.L_x, net.L_d) == (D, W, skip, 10, 4)
        blob = ops.pack_module(sd, "model_coarse.", net)
        assert blob.numel() == _lib.lib().mi_nerf_packed_bytes(ctypes.byref(net))


def test_pack_rejects_bad_shapes_and_unsupported_nets():
    sd = synthetic.make_state_dict(1, 4, 128)
    net = weights.infer_net(sd)
    bad = dict(sd); bad["model_coarse.linear_feat.weight"] = np.zeros((128, 64), np.float32)
    with pytest.raises(_lib.MiNerfError):
        ops.pack_module(bad, "model_coarse.", net)
    with pytest.raises(_lib.MiNerfError, match="unsupported width W=600"):          # wider than the widest kernel: refused, not truncated
        ops.pack_module(synthetic.make_state_dict(1, 4, 600), "model_coarse.", ops.make_net(4, 600, -1))
    narrow = synthetic.make_state_dict(1, 4, 64)                                    # narrower: fp32 inference pads (below); the other blobs refuse
    for kw in (dict(bf16=True), dict(f16s=True), dict(backward=True)):
        with pytest.raises(_lib.MiNerfError):
            ops.pack_module(narrow, "model_coarse.", ops.make_net(4, 64, -1), **kw)
    with pytest.raises(_lib.MiNerfError, match="training kernels exist for W = 128 and 256"):
        ops.train_layout(ops.make_net(4, 64, -1), 8, 32)


def test_cpu_tensors_are_refused():
    with pytest.raises(_lib.MiNerfError):
        ops.stratified_z(2.0, 6.0, torch.rand(4, 64))


# widths the reference's --netWidth accepts (config.py:57) but no kernel is instantiated for: packed into the next kernel width with
# zero weights for the units the network does not have (layout.h kernel_width) -- 64 -> 128, 200 -> 256, odd halves (W // 2), tiny
@pytest.mark.parametrize("D,W,skip", [(8, 256, 4), (4, 128, 4), (4, 64, 4), (6, 64, 2), (5, 200, 1), (3, 100, 0), (4, 31, 4), (2, 2, -1), (8, 129, 4)])
def test_blob_emulation_matches_oracle_embedded(D, W, skip):
    sd = synthetic.make_state_dict(11, D, W, skips=() if skip < 0 else (skip,))
    net = weights.infer_net(sd)
    assert net.W == W
    blob = ops.pack_module(sd, "model_fine.", net).numpy()
    assert blob.nbytes == _lib.lib().mi_nerf_packed_bytes(ctypes.byref(ops.make_net(D, 128 if W <= 128 else 256, skip)))      # the padded width's layout
    rs = np.random.RandomState(0)
    x = rs.uniform(-1, 1, size=(32, 90)).astype(np.float32)
    emu = Emu(blob)
    # embedded-mode registers: gather channels exactly as gather_regs<> does
    def gather(row_block, L):
        K = ((3 * L + 2 + 3) // 4) * 4
        reg = np.zeros((K, 64))
        for s in range(3 * L):
            ch = 3 + 6 * (s // 3) + (s % 3)
            reg[s, :32], reg[s, 32:] = row_block[:, ch], row_block[:, ch + 3]
        reg[3 * L, :32], reg[3 * L, 32:] = row_block[:, 0], row_block[:, 1]
        reg[3 * L + 1, :32] = row_block[:, 2]
        return reg
    out = emu.tile(gather(x[:, :63].astype(np.float64), 10), de=gather(x[:, 63:].astype(np.float64), 4))
    ref = R.mlp_forward(sd, "model_fine.", torch.from_numpy(x), D, 63, 27, skips=() if skip < 0 else (skip,), dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


@pytest.mark.parametrize("D,W,skip", [(3, 512, 0), (2, 300, -1), (4, 257, 1), (3, 384, 1), (2, 400, 0)])
def test_wide_blob_emulation_matches_oracle(D, W, skip):
    """Networks wider than 256 (--netWidth 512, config.py:57): packed for the 512-wide kernel in the W16 stream order (mlp_fp32_wide.hip:
    16 points per wave on v_mfma_f32_16x16x4_f32); tests/blob_emulator.py EmuWide walks the stream as that kernel does.  Embedded mode (the
    direction k-steps in the stream) and fused mode (hoisted direction bias) against the oracle in fp64."""
    from tests.blob_emulator import EmuWide
    skips = () if skip < 0 else (skip,)
    sd = synthetic.make_state_dict(19, D, W, skips=skips)
    net = weights.infer_net(sd)
    blob = ops.pack_module(sd, "model_fine.", net).numpy()
    assert blob.nbytes == _lib.lib().mi_nerf_packed_bytes(ctypes.byref(ops.make_net(D, 384 if W <= 384 else 512, skip)))
    emu = EmuWide(blob)
    rs = np.random.RandomState(2)
    x = rs.uniform(-1, 1, size=(16, 90))
    out = emu.tile(emu.gather_regs(10, x[:, :63]), de=emu.gather_regs(4, x[:, 63:]))
    ref = R.mlp_forward(sd, "model_fine.", torch.from_numpy(x), D, 63, 27, skips=skips, dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)
    ray = rs.normal(size=6)
    z = np.sort(rs.uniform(2, 6, 16))
    p = ray[:3, None] + ray[3:, None] * z[None, :]
    v = ray[3:] / np.linalg.norm(ray[3:])
    g = R.posenc(torch.from_numpy(v[None]), 4)[0].numpy()
    out = emu.tile(emu.enc_regs(10, p), dir_gamma=g)
    xx = torch.cat([R.posenc(torch.from_numpy(p.T.copy()), 10), torch.from_numpy(g)[None].expand(16, 27)], -1)
    ref = R.mlp_forward(sd, "model_fine.", xx, D, 63, 27, skips=skips, dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


@pytest.mark.parametrize("D,W", [(8, 256), (8, 64), (8, 192), (4, 100)])
def test_blob_emulation_matches_oracle_fused(D, W):
    sd = synthetic.make_state_dict(5, D, W)
    net = weights.infer_net(sd)
    blob = ops.pack_module(sd, "model_coarse.", net).numpy()
    rs = np.random.RandomState(1)
    ray = rs.normal(size=6)
    z = np.sort(rs.uniform(2, 6, 32))
    p = ray[:3, None] + ray[3:, None] * z[None, :]
    v = ray[3:] / np.linalg.norm(ray[3:])
    emu = Emu(blob)
    g = R.posenc(torch.from_numpy(v[None]), 4)[0].numpy()
    out = emu.tile(emu.enc_regs(10, p), dir_gamma=g)
    x = torch.cat([R.posenc(torch.from_numpy(p.T.copy()), 10), torch.from_numpy(g)[None].expand(32, 27)], -1)
    ref = R.mlp_forward(sd, "model_coarse.", x, D, 63, 27, dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


def test_model_mirror_has_reference_checkpoint_keys():
    """The NeRF mirror must accept the reference's checkpoints: same state_dict keys and shapes as model/NeRF.py:24-30,58-59."""
    from nerf_pytorch_paeng_amd.model import NeRF
    for D, W in ((8, 256), (4, 128)):
        m = NeRF(D, W, 63, 27, skips=[4], gt_camera_param=(None, None))
        sd = synthetic.make_state_dict(0, D, W)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        want = {k: tuple(v.shape) for k, v in sd.items()}
        assert got == want
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        assert m.get_camera_gt() == (None, None)
    with pytest.raises(_lib.MiNerfError):        # parameters on CPU: the MI355X path refuses, it never falls back
        m(torch.rand(2, 90))


# ---------------------------------------------------------------------------------------------------
# training path: backward-data blob, flat parameter order, device-pack maps
# ---------------------------------------------------------------------------------------------------
def _regs_from_rows(X):
    """[32 points, F] -> [F/2, 64] per-lane B registers in accumulator layout"""
    from tests.blob_emulator import COL, HH, _row_of
    F = X.shape[1]
    reg = np.zeros((F // 2, 64))
    for s in range(F // 2):
        reg[s] = X[COL, 32 * (s // 16) + _row_of(s % 16, HH)]
    return reg


def _rows_from_acc(acc, NT):
    from tests.blob_emulator import COL, HH, _row_of
    X = np.zeros((32, 32 * NT))
    for t in range(NT):
        for r in range(16):
            X[COL, 32 * t + _row_of(r, HH)] = acc[t, r]
    return X


@pytest.mark.parametrize("D,W,skip", [(8, 256, 4), (4, 128, 1)])
def test_backward_blob_emulation_is_the_transposed_chain(D, W, skip):
    """Walk the backward-data stream with MFMA semantics (no ReLU masks: a pure layout check) and compare with
    delta @ W[:, h-block] applied in the order mlp_dgrad_kernel consumes the parts."""
    sd = synthetic.make_state_dict(21, D, W, skips=(skip,))
    net = ops.make_net(D, W, skip)
    prefix = "model_coarse."
    blob = ops.pack_module(sd, prefix, net, backward=True).numpy()
    hdr = np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)
    assert hdr[0] == 0x4D494E46 and hdr[1] == 2 and hdr[7] == 1024 and hdr[8] == blob.size - 1024
    emu = Emu.__new__(Emu)
    emu.stream = np.frombuffer(blob[1024:].tobytes(), dtype=np.float32).astype(np.float64)
    emu.pos = 0
    NT = W // 32
    rs = np.random.RandomState(1)
    delta = rs.normal(size=(32, W // 2))
    w = lambda k: np.asarray(sd[prefix + k], dtype=np.float64)
    chain = [w("linear_d.weight")[:, :W], w("linear_feat.weight")]
    for l in range(D - 1, 0, -1):
        wl = w(f"linear_x.{l}.weight")
        chain.append(wl[:, 63:] if l == skip + 1 else wl)
    for Wm in chain:
        acc = emu.gemm_part(np.zeros((NT, 16, 64)), NT, _regs_from_rows(delta))
        got = _rows_from_acc(acc, NT)
        want = delta @ Wm
        np.testing.assert_allclose(got, want, atol=1e-9 * np.abs(want).max(), rtol=1e-9)
        delta = want
    assert emu.pos * 1024 == blob.size - 1024


def test_flat_parameter_order_and_pack_maps():
    net = ops.make_net(8, 256, 4)
    sd = synthetic.make_state_dict(4, 8, 256)
    from nerf_pytorch_paeng_amd.model import NeRF
    model = NeRF(8, 256, 63, 27)
    names = [n for n, _ in model.model_coarse.named_parameters()]
    assert names == ops.param_names(net)
    assert sum(p.numel() for p in model.model_coarse.parameters()) == ops.param_count(net)
    flat = ops.flatten_params(sd, "model_fine.", net)
    for backward in (False, True):
        mp = ops.pack_map(net, backward)
        host = ops.pack_module(sd, "model_fine.", net, backward=backward).view(torch.float32)
        out = torch.zeros(mp.numel())
        nz = mp > 0
        out[nz] = flat[(mp[nz] - 1).long()]
        assert torch.equal(out[256:], host[256:])
        assert int(mp.min()) == 0 and int(mp.max()) <= flat.numel()


# ---------------------------------------------------------------------------------------------------
# bf16 variant: output-tile-major stream, heads on the matrix pipe
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,skip", [(8, 4), (3, -1), (4, 1), (7, 5)])
def test_bf16_blob_emulation_matches_bf16_oracle(D, skip):
    """The bf16 packer + the kernel's operand bookkeeping (tests/blob_emulator_bf16.py walks the stream as mlp_bf16.hip does) against
    the oracle with the same rounding points (R.mlp_forward_bf16), both accumulating in fp64."""
    from tests.blob_emulator_bf16 import EmuBf16
    W = 256
    sd = synthetic.make_state_dict(5, D, W, skips=(skip,) if skip >= 0 else ())
    net = weights.infer_net(sd)
    assert net.skip == (skip if skip + 1 < D else skip) or skip < 0
    blob = ops.pack_module(sd, "model_coarse.", net, bf16=True).numpy()
    rs = np.random.RandomState(2)
    ray = rs.normal(size=6)
    z = np.sort(rs.uniform(2, 6, 16))
    p = (ray[:3, None] + ray[3:, None] * z[None, :]).T.astype(np.float32)           # [16, 3]: one MFMA point tile
    v = ray[3:] / np.linalg.norm(ray[3:])
    g = R.posenc(torch.from_numpy(v[None]), 4)[0].numpy()
    emu = EmuBf16(blob)
    out = emu.tile(p.astype(np.float64), g)
    x = torch.cat([R.posenc(torch.from_numpy(p).double(), 10), torch.from_numpy(g)[None].expand(16, 27)], -1)
    ref = R.mlp_forward_bf16(sd, "model_coarse.", x, D, 63, 27, skips=(skip,) if skip >= 0 else (), dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


@pytest.mark.parametrize("D,skip", [(8, 4), (4, -1)])
def test_bf16_pack_map_reproduces_the_host_blob(D, skip):
    """The gather map of the device-side bf16 packer (mi_nerf_pack_map_bf16), applied here in numpy with the same rounding, must
    give the host packer's blob bit for bit: stream elements rounded to bf16, fp32 side tables."""
    from tests.blob_emulator_bf16 import bf16_round
    net = ops.make_net(D, 256, skip)
    sd = synthetic.make_state_dict(11, D, 256, skips=(skip,) if skip >= 0 else ())
    blob = ops.pack_module(sd, "model_fine.", net, bf16=True).numpy()
    hdr = np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)
    so, sb, sdo, sf = int(hdr[7]), int(hdr[8]), int(hdr[10]), int(hdr[11])
    m = ops.pack_map_bf16(net).numpy()
    assert m.shape[0] == sb // 2 + sf
    flat = ops.flatten_params(sd, "model_fine.", net).numpy()
    src = np.where(m > 0, flat[np.maximum(m, 1) - 1], np.float32(0.0)).astype(np.float32)
    want_stream = np.frombuffer(blob[so:so + sb].tobytes(), dtype=np.uint16)
    got_stream = (bf16_round(src[:sb // 2]).view(np.uint32) >> 16).astype(np.uint16)
    assert np.array_equal(got_stream, want_stream)
    want_side = np.frombuffer(blob[sdo:sdo + 4 * sf].tobytes(), dtype=np.float32)
    assert np.array_equal(src[sb // 2:], want_side)


@pytest.mark.parametrize("src,mfma_name,min_mfma", [("mlp_bf16.hip", "v_mfma_f32_16x16x32_bf16", 4000), ("mlp_f16s.hip", "v_mfma_f32_16x16x32_f16", 6000),
                                                   ("mlp_f16s_stash.hip", "v_mfma_f32_16x16x32_f16", 6000), ("dgrad_f16s.hip", "v_mfma_f32_16x16x32_f16", 3000)])
def test_fragment_file_kernels_own_m0_and_the_agpr_file(src, mfma_name, min_mfma):
    """mlp_bf16.hip and the split-precision kernels (mlp_f16s.hip, mlp_f16s_stash.hip, dgrad_f16s.hip) set M0 without saving it and address the whole AGPR file by explicit register numbers: both are only
    sound while hipcc itself never touches M0 / an AGPR in those kernels.  Disassemble the objects and check."""
    import os
    import re
    import subprocess
    from nerf_pytorch_paeng_amd import build
    obj = build.ensure_object(src)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    import shutil
    import tempfile
    work = tempfile.mkdtemp()
    try:
        local = os.path.join(work, "k.o")
        shutil.copy(obj, local)
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=work)      # extracts k.o.0.hipv4-...-gfx950
        dev = [f for f in os.listdir(work) if f.endswith("gfx950")]
        assert len(dev) == 1, os.listdir(work)
        asm = subprocess.run([objdump, "-d", os.path.join(work, dev[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(work)
    lines = [l.split("//")[0].strip() for l in asm.splitlines()]
    mfma = [l for l in lines if l.startswith(mfma_name)]
    assert len(mfma) > min_mfma
    assert not any(l.startswith("v_accvgpr_read") for l in lines)                 # the compiler never moves data out of the file
    assert not any(l.startswith("scratch_") for l in lines)                       # no spills
    m0 = [l for l in lines if re.search(r"\bm0\b", l)]
    assert m0 and all(re.fullmatch(r"s_mov_b32 m0, s\d+", l) for l in m0), m0[:5]  # only our "s_mov_b32 m0, sN"
    for l in lines:                                                               # AGPRs appear only as MFMA B operands / accvgpr_write targets
        if re.search(r"\ba\[?\d", l):
            assert l.startswith("v_accvgpr_write_b32 a") or l.startswith(mfma_name + " v["), l


@pytest.mark.parametrize("src,min_mfma", [("mlp_bf16.hip", 4000), ("mlp_f16s.hip", 6000), ("mlp_f16s_stash.hip", 6000), ("dgrad_f16s.hip", 3000),
                                          ("mlp_fp32.hip", 800), ("mlp_fp32_wide.hip", 10000), ("mlp_train.hip", 60)])
def test_mfma_destinations_and_c_operands_are_left_alone_for_their_wait_states(src, min_mfma):
    """An MFMA written as an asm statement gets none of hipcc's software wait states: nothing may read or write its destination tuple within
    P + 4 wait states of its issue, no VALU may write its C operand within 7 / 13 (ISA guide 4.5; cdna_hip_programming.md 5.7 item 2).  Round 3's
    NOPACK ablation build of the f16s kernel page-faulted on exactly this (dead accumulators -> their registers reused for the DMA address and
    the ring's fetch offset -> overwritten by the late matrix write), and the audit that followed found two near misses in shipped kernels
    (DESIGN 3.4).  tools/mfma_hazard_check.py walks the disassembly; the builtin-MFMA kernels (fp32, training) ride along as a check of the checker."""
    import os
    import sys
    from nerf_pytorch_paeng_amd import build
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mfma_hazard_check as H
    obj = build.ensure_object(src)
    if not os.path.exists(H.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    seen = 0
    for name, ins in H.kernels_of(H.device_asm(obj)).items():
        n, bad = H.check(ins)
        seen = max(seen, n)
        assert not bad, f"{name}: {len(bad)} pairs, first: {bad[:3]}"
    assert seen > min_mfma


# Pairs per kernel that are inside their window when an intervening MFMA is priced at ONE wait state (what LLVM's recogniser assumes) and
# outside it at the pipe's ISSUE INTERVAL (P / 2 = 4 wait states for the 16x16x32 f16 / bf16 shapes: tools/mfma_probe.hip measured one
# 8-pass MFMA per 16 clocks per wave, which is also what the 2.5 PFLOP/s peak is made of): the reliance of the asm kernels' schedules on
# the matrix pipe's rate, as of hipcc 7.2 / round 5.  The in-order wave cannot issue the instruction behind an MFMA earlier than that
# interval, so these are safe; but the NUMBER must not grow silently with a compiler change -- it is pinned here.
STRICT_RELIANCE = {
    ("mlp_bf16.hip", "mlp_bf16_kernelILi256ELi10ELi4ELi4ELi0ELi4E"): 4,
    ("mlp_bf16.hip", "mlp_bf16_kernelILi256ELi10ELi4ELi2ELi0ELi4E"): 195,
    ("mlp_bf16.hip", "mlp_bf16_kernelILi256ELi10ELi4ELi4ELi2ELi4E"): 199,
    ("mlp_f16s.hip", "mlp_f16s_kernelILb0E"): 74,
    ("mlp_f16s_stash.hip", "mlp_f16s_kernelILb1E"): 13,
    ("dgrad_f16s.hip", "dgrad_f16s_kernel"): 54,
}


@pytest.mark.parametrize("src", ["mlp_bf16.hip", "mlp_f16s.hip", "mlp_f16s_stash.hip", "dgrad_f16s.hip", "mlp_fp32.hip", "mlp_fp32_wide.hip", "mlp_train.hip"])
def test_reliance_on_the_mfma_issue_interval_does_not_grow(src, capsys):
    """`mfma_hazard_check.py --strict` per kernel: printed, and bounded by today's counts (builtin-MFMA kernels: zero -- hipcc pads those itself)."""
    import os
    import sys
    from nerf_pytorch_paeng_amd import build
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mfma_hazard_check as H
    if not os.path.exists(H.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    seen = set()
    for name, ins in H.kernels_of(H.device_asm(build.ensure_object(src))).items():
        n, bad = H.check(ins, strict=True)
        if n == 0:
            continue
        keys = [k for k in STRICT_RELIANCE if k[0] == src and k[1] in name]
        limit = STRICT_RELIANCE[keys[0]] if keys else 0
        seen.update(keys)
        with capsys.disabled():
            print(f"\n  strict  {src}  {name[:84]}: {n} MFMAs, {len(bad)} pairs rely on the issue interval (pinned: {limit})")
            for b in sorted(bad, key=lambda x: int(x.split("+")[1].split("ws")[0]))[:3]:
                print("      ", b[:200])
        assert len(bad) <= limit, f"{name}: {len(bad)} pairs rely on the MFMA issue interval, {limit} when this was pinned; worst: {bad[:3]}"
    assert seen == {k for k in STRICT_RELIANCE if k[0] == src}, (src, seen)          # every pinned kernel is still in the object


def test_hazard_checker_skips_host_only_objects():
    import os
    import sys
    from nerf_pytorch_paeng_amd import build
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mfma_hazard_check as H
    if not os.path.exists(H.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    obj = build.ensure_object("pack.cpp")
    assert H.device_asm(obj) == "" and H.main(["x", obj]) == 0


def test_no_ablation_switches_in_the_shipped_sources():
    """Round 3's only GPU fault came from an A/B build of the kernel sources.  The timing switches are gone (tools/ABLATIONS.md): the shipped
    flags define no MN_* macro, and every MN_* conditional left in csrc/ (today: MN_DIAG) is named by a script under tools/ that builds it."""
    import os
    import re
    from nerf_pytorch_paeng_amd import build
    flags = [*build.FLAGS, *[f for v in build.FILE_FLAGS.values() for f in v]]
    assert not [f for f in flags if f.startswith("-DMN_")], flags
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools_text = "".join(open(os.path.join(root, "tools", f), errors="ignore").read() for f in os.listdir(os.path.join(root, "tools"))
                         if f.endswith((".py", ".sh")))
    build_text = open(os.path.join(root, "nerf_pytorch_paeng_amd", "build.py")).read()
    conds, n_if = set(), 0
    for f in sorted(os.listdir(build.CSRC)):
        for line in open(os.path.join(build.CSRC, f)):
            if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", line):
                n_if += 1
                conds.update(re.findall(r"MN_[A-Z0-9_]+", line))
    assert conds == {"MN_DIAG"}, conds
    for c in conds:
        assert f"-D{c}" in tools_text + build_text, c
    assert n_if <= 14, n_if                                           # 12 MN_DIAG sites today (30 conditionals before the prune)


def test_mfma_hazard_checker_sees_a_planted_hazard():
    """The checker on hand-made listings: the NOPACK build's two faulting patterns, a load-return write, a clean stream."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mfma_hazard_check as H

    def listing(*body):
        return H.kernels_of("0000000000001900 <k>:\n" + "".join(f"\t{b}    // 00: 00\n" for b in body))["k"]
    mf = "v_mfma_f32_16x16x32_f16 v[54:57], v[80:83], a[120:123], v[54:57]"
    # 1. the DMA address pair computed into a pending destination
    n, bad = H.check(listing(mf, "ds_read_b128 v[0:3], v72", "s_nop 0", "v_lshl_add_u64 v[54:55], v[66:67], 0, s[10:11]", "global_load_lds_dwordx4 v[54:55], off"))
    assert n == 1 and any(b.startswith("WAW") for b in bad) and any(b.startswith("RAW") for b in bad)
    # 2. the ring's fetch offset behind a barrier: instruction counts, not time, are what the check sees
    n, bad = H.check(listing("v_mfma_f32_16x16x32_f16 v[12:15], v[44:47], v[56:59], v[12:15]", "s_waitcnt vmcnt(8)", "s_barrier", "v_add_u32_e32 v12, 0x8000, v74"))
    assert len(bad) == 1 and bad[0].startswith("WAW")
    # 3. a shuffle result landing in the unread element of the colour tile (the shipped STASH kernel until round 4)
    n, bad = H.check(listing("v_mfma_f32_16x16x32_f16 v[76:79], v[104:107], a[88:91], v[76:79]", "ds_bpermute_b32 v79, v188, v85"))
    assert len(bad) == 1
    # 4. VALU write of a C operand right behind the MFMA (the shipped bf16 32-point shape until round 4); legal after 7 wait states
    c = "v_mfma_f32_16x16x32_bf16 v[22:25], v[36:39], a[32:35], v[14:17]"
    assert H.check(listing(c, "v_add_u32_e32 v14, 0x8000, v54"))[1]
    assert not H.check(listing(c, "s_nop 6", "v_add_u32_e32 v14, 0x8000, v54"))[1]
    # 5. far enough: 12 wait states; an accumulate chain through C needs none; intervening MFMAs count their issue interval (4)
    assert not H.check(listing(mf, "s_nop 11", "v_mov_b32_e32 v54, v1"))[1]
    assert H.check(listing(mf, "s_nop 10", "v_mov_b32_e32 v54, v1"))[1]
    assert not H.check(listing(mf, mf, mf))[1]
    other = "v_mfma_f32_16x16x32_f16 v[0:3], v[80:83], a[0:3], v[0:3]"
    assert not H.check(listing(mf, other, other, other, "v_mov_b32_e32 v9, v54"))[1]
    assert H.check(listing(mf, other, other, "v_mov_b32_e32 v9, v54"))[1]
    assert H.check(listing(mf, other, other, other, "v_mov_b32_e32 v9, v54"), strict=True)[1]
    # 6. the builtin-MFMA kernels' accumulators live in AGPRs and are read out by explicit v_accvgpr_read statements: an f32-input 32x32x2
    #    (16 passes, not on the XDL pipe) needs P + 2 = 18 wait states -- hipcc's own `s_nop 15; s_nop 1`
    f32 = "v_mfma_f32_32x32x2_f32 a[32:47], v27, v131, a[32:47]"
    assert not H.check(listing(f32, "s_nop 15", "s_nop 1", "v_accvgpr_read_b32 v32, a32"))[1]
    assert H.check(listing(f32, "s_nop 15", "s_nop 0", "v_accvgpr_read_b32 v32, a32"))[1]
    f32b = "v_mfma_f32_32x32x2_f32 a[48:63], v27, v131, a[48:63]"
    assert not H.check(listing(f32, f32b, "s_nop 1", "v_accvgpr_read_b32 v32, a32"))[1]         # another tile's MFMA holds the wave 16 wait states
    assert H.check(listing(f32, f32b, "s_nop 0", "v_accvgpr_read_b32 v32, a32"))[1]
    # 7. a packed fragment written into the AGPR file is an MFMA operand two wait states later at the earliest
    wr = "v_accvgpr_write_b32 a120, v5"
    assert H.check(listing(wr, mf))[1] and H.check(listing(wr, "s_nop 0", mf))[1] and not H.check(listing(wr, "s_nop 1", mf))[1]


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("D,skip", [(8, 4), (3, 0), (2, -1)])
def test_f16s_blobs_reconstruct_the_weights_through_their_gather_maps(D, skip, backward):
    """Split-precision blobs (forward stream + side tables; transposed backward stream): every stream element named by the gather map holds
    the hi (first 512 halves of a 1024-half pair block) or lo (second 512) half of that weight, hi + lo 2^-11 == w to 2^-21 relative; the
    maps cover every weight the chain uses; side-table floats are copies."""
    from nerf_pytorch_paeng_amd import synthetic, weights
    sd = synthetic.make_state_dict(17, D, 256, skips=() if skip < 0 else (skip,))
    net = weights.infer_net(sd)
    blob = ops.pack_module(sd, "model_fine.", net, backward=backward, f16s=True).numpy()
    m = ops.pack_map_f16s(net, backward=backward).numpy()
    flat = np.concatenate([np.asarray(sd["model_fine." + k], dtype=np.float32).reshape(-1) for k in ops.param_names(net)])
    hdr = blob[:64].view(np.uint32)
    stream_off, stream_bytes = int(hdr[7]), int(hdr[8])
    n_stream = stream_bytes // 2
    st = blob[stream_off:stream_off + stream_bytes].view(np.float16).astype(np.float64)
    assert m.size >= n_stream and n_stream % 1024 == 0
    mm = m[:n_stream].reshape(-1, 2, 512)
    assert np.array_equal(mm[:, 0], mm[:, 1])                                  # a pair block: hi and lo halves of the SAME weights
    v = st.reshape(-1, 2, 512)
    rec = v[:, 0] + v[:, 1] / 2048.0
    used = mm[:, 0] > 0
    want = flat[mm[:, 0][used] - 1].astype(np.float64)
    assert np.all(np.abs(rec[used] - want) <= 2.0 ** -21 * np.abs(want) + 2.0 ** -35)     # below the f16 normal range (2^-14) the error is absolute
    assert np.all(rec[~used] == 0.0)
    if not backward:
        side_off, side_floats = int(hdr[10]), int(hdr[11])
        side = blob[side_off:side_off + 4 * side_floats].view(np.float32)
        ms = m[n_stream:n_stream + side_floats]
        assert np.array_equal(side[ms > 0], flat[ms[ms > 0] - 1]) and np.all(side[ms == 0] == 0.0)
    # coverage: the forward blob names every parameter except none; the backward stream every trunk weight that multiplies an activation
    names = ops.param_names(net)
    sizes = [int(np.prod(sd["model_fine." + k].shape)) for k in names]
    offs = np.cumsum([0] + sizes)
    hit = np.zeros(flat.size, dtype=bool)
    hit[m[m > 0] - 1] = True
    for k, o, sz in zip(names, offs[:-1], sizes):
        frac = hit[o:o + sz].mean()
        if not backward:
            assert frac == 1.0, (k, frac)
        elif k == "linear_feat.weight" or (k.startswith("linear_x.") and k.endswith(".weight") and not k.startswith("linear_x.0.")):
            assert frac >= 256.0 / (256.0 + 63.0) - 1e-6, (k, frac)            # the skip layer's gamma(x) block has no gradient path


@pytest.mark.parametrize("D,W,Wk,skip", [(4, 64, 128, 1), (8, 200, 256, 4), (3, 31, 128, 0), (5, 100, 256, -1), (2, 128, 256, -1)])
def test_padding_a_network_into_a_wider_one_keeps_its_function(D, W, Wk, skip):
    """weights.pad_index_map / padded_state_dict (the training path's and the bf16 / split-precision packers' way to run a width without kernels
    of its own): the scattered parameters describe a Wk-wide network that computes the SAME function (oracle, fp64: the extra units are exact
    zeros), the map is injective, and gathering through it inverts the scatter."""
    skips = () if skip < 0 else (skip,)
    sd = synthetic.make_state_dict(23, D, W, skips=skips)
    net = weights.infer_net(sd)
    wide = ops.make_net(D, Wk, net.skip, net.L_x, net.L_d)
    idx = weights.pad_index_map(net, wide)
    assert idx.numel() == ops.param_count(net) and idx.unique().numel() == idx.numel() and int(idx.max()) < ops.param_count(wide)
    big = weights.padded_state_dict(sd, "model_fine.", net, wide)
    assert weights.infer_net({k.replace("model_fine.", "model_coarse."): v for k, v in big.items()}).W == Wk
    flat = np.concatenate([sd["model_fine." + k].reshape(-1) for k in ops.param_names(net)])
    flat_big = np.concatenate([big["model_fine." + k].reshape(-1) for k in ops.param_names(wide)])
    assert np.array_equal(flat_big[idx.numpy()], flat) and np.count_nonzero(flat_big) == np.count_nonzero(flat)
    x = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, size=(50, 90)))
    a = R.mlp_forward(sd, "model_fine.", x, D, 63, 27, skips=skips, dtype=torch.float64)
    b = R.mlp_forward(big, "model_fine.", x, D, 63, 27, skips=skips, dtype=torch.float64)
    np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-12, atol=1e-13)      # exact zeros added; torch's GEMM blocks the longer sums differently (last fp64 bits)
