"""Error behaviour of the C ABI ("returning int, 0 = ok, non-zero = HIP / argument error; no exceptions across the ABI, error
text via a last_error() accessor"), checked WITHOUT a GPU: every entry point that takes a pointer is called with plausible scalars and NULL
pointers and must answer MI_NERF_EINVAL (1) with a message -- which also proves that the NULL / size checks sit in front of the first HIP
call (on this box a call that got as far as a launch answers MI_NERF_EHIP, "no ROCm-capable device").  No call here can reach a kernel."""
import ctypes as C

import pytest

from nerf_pytorch_paeng_amd import _lib, ops

EINVAL, EHIP = 1, 2
# scalar arguments (by position) that must be valid for the NULL check to be the one that fires
OVERRIDES = {
    "mi_nerf_composite": {3: 6},                       # ray_stride
    "mi_nerf_composite_backward": {3: 6},
    "mi_nerf_render_rays": {9: 1 << 30},               # workspace_bytes
    "mi_nerf_pack_map": {1: 0, 3: 1 << 24},            # kind, map_len
    "mi_nerf_mlp_rays_train": {8: 1 << 30},            # stash_bytes
    "mi_nerf_mlp_rays_train_f16s": {8: 1 << 30},
    "mi_nerf_mlp_embedded_train": {6: 1 << 30},
    "mi_nerf_mlp_backward": {10: 1 << 34},             # work_bytes
    "mi_nerf_mlp_backward_mode": {10: 1 << 34, 12: 0, 13: 0},     # work_bytes, stage, mode
    "mi_nerf_mlp_embedded_backward": {8: 1 << 34},
    "mi_nerf_rays_rgb": {0: 8, 1: 8, 5: 2},            # W, H, n_img
    "mi_nerf_time_mlp_rays": {7: 1, 8: 0},             # iters, mode
    "mi_nerf_fill_uniform": {4: 8},
    "mi_nerf_wgrad_products": {0: 1},
    "mi_nerf_wgrad_products_f16s": {0: 1},
}


def _pointer_entries():
    out = []
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        if restype is C.c_int and any(a not in (C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_size_t) for a in argtypes):
            out.append(name)
    return out


def _args(name, net, cfg):
    args = []
    for i, a in enumerate(_lib.SIGNATURES[name][1]):
        if i in OVERRIDES.get(name, {}):
            args.append(OVERRIDES[name][i])
        elif a is _lib._NETP:
            args.append(C.byref(net))
        elif a is _lib._CFGP:
            args.append(C.byref(cfg))
        elif a in (C.c_int, C.c_int64, C.c_uint32):
            args.append(4)
        elif a is C.c_size_t:
            args.append(0)
        elif a is C.c_float:
            args.append(1.0)
        else:
            args.append(None)                              # every data pointer NULL
    return args


@pytest.mark.parametrize("name", _pointer_entries())
def test_null_pointers_are_refused_before_any_hip_call(name):
    if name == "mi_nerf_selftest_mfma":
        pytest.skip("takes only a stream: nothing to refuse")
    lib = _lib.lib()
    net, cfg = ops.make_net(8, 256, 4), ops.render_cfg(2.0, 6.0, 64, 128, False, False)
    rc = getattr(lib, name)(*_args(name, net, cfg))
    msg = lib.mi_nerf_last_error().decode()
    assert rc == EINVAL, (name, rc, msg)
    assert msg and "HIP error" not in msg, (name, msg)


@pytest.mark.parametrize("name", [n for n in _pointer_entries() if _lib._NETP in _lib.SIGNATURES[n][1]])
def test_null_net_and_unsupported_shapes_are_refused(name):
    lib = _lib.lib()
    cfg = ops.render_cfg(2.0, 6.0, 64, 128, False, False)
    good = ops.make_net(8, 256, 4)
    a = _args(name, good, cfg)
    a[list(_lib.SIGNATURES[name][1]).index(_lib._NETP)] = None
    assert getattr(lib, name)(*a) == EINVAL and lib.mi_nerf_last_error()
    for bad in (ops.make_net(8, 600, 4), ops.make_net(8, 1, 4), ops.make_net(0, 256, -1), ops.make_net(40, 256, 4)):   # widths / depths no kernel can run (2 <= W <= 512 pads)
        b = _args(name, bad, cfg)
        assert getattr(lib, name)(*b) == EINVAL, (name, bad.D, bad.W)


def test_tile_gather_helpers_refuse_bad_arguments_before_touching_rccl():
    """mi_nerf_comm_* / mi_nerf_all_gather_tiles (SURVEY.md 8(b) "thin RCCL helpers"): argument errors are MI_NERF_EINVAL with a message, and are
    found before librccl is even looked for (this box has no GPU; an RCCL or HIP call would answer 3 or 2)."""
    lib = _lib.lib()
    ident = C.create_string_buffer(128)
    handle = C.c_void_p()
    for world, rank in ((0, 0), (2, 2), (2, -1)):
        assert lib.mi_nerf_comm_init_rank(ident, world, rank, C.byref(handle)) == EINVAL and handle.value is None
        assert b"rank" in lib.mi_nerf_last_error()
    assert lib.mi_nerf_comm_init_rank(None, 2, 0, C.byref(handle)) == EINVAL
    fake = C.create_string_buffer(64)                                   # not a handle of mi_nerf_comm_init_rank
    assert lib.mi_nerf_all_gather_tiles(fake, None, 1, 8, 8, 4, None, None, 0, None) == EINVAL and b"handle" in lib.mi_nerf_last_error()
    assert lib.mi_nerf_comm_destroy(fake) == EINVAL and lib.mi_nerf_comm_info(fake, None, None, None) == EINVAL
    # a handle is valid iff it is in the library's registry: an address that is not even mapped (or a handle destroyed earlier, whose memory
    # is gone) is refused WITHOUT being read -- up to round 5 the check read a magic word through the pointer
    wild = C.c_void_p(0x10)
    assert lib.mi_nerf_comm_destroy(wild) == EINVAL and lib.mi_nerf_comm_info(wild, None, None, None) == EINVAL
    assert lib.mi_nerf_all_gather_tiles(wild, None, 1, 8, 8, 4, None, None, 0, None) == EINVAL and b"handle" in lib.mi_nerf_last_error()
    # staging: nothing for equal blocks; world x largest block for a ragged split (fern: 378 rows over 8 ranks -> 8 x 48 rows)
    assert lib.mi_nerf_all_gather_staging_bytes(8, 800, 800, 4) == 0
    assert lib.mi_nerf_all_gather_staging_bytes(8, 378, 504, 4) == 8 * 48 * 504 * 4 * 4
    assert lib.mi_nerf_all_gather_staging_bytes(1, 378, 504, 4) == 0
    assert lib.mi_nerf_all_gather_staging_bytes(8, 4096, 1, 4) == 0 and lib.mi_nerf_all_gather_staging_bytes(3, 4096, 1, 4) == 3 * 1366 * 16
    assert lib.mi_nerf_all_gather_staging_bytes(0, 8, 8, 4) == 0 and lib.mi_nerf_all_gather_staging_bytes(16, 8, 8, 4) == 0      # refused geometry
    assert lib.mi_nerf_unpad_tiles(fake, 16, 8, 8, 4, fake, None) == EINVAL and b"cannot be split" in lib.mi_nerf_last_error()
    # librccl itself: resolved at first use (this image has ROCm's under /opt/rocm/lib, and torch's when torch.distributed is loaded); when it
    # cannot be, the answer is MI_NERF_ERCCL (3) with the loader's text, never a crash -- and nothing above needed it
    rc = lib.mi_nerf_rccl_available()
    assert rc in (0, 3) and (rc == 0 or b"RCCL is not available" in lib.mi_nerf_last_error())


def test_the_error_text_is_per_thread():
    """mi_nerf_last_error() is thread-local: a failure on another thread does not overwrite this thread's message."""
    import threading
    lib = _lib.lib()
    net = ops.make_net(8, 256, 4)
    assert lib.mi_nerf_mlp_rays(C.byref(net), None, None, None, -5, 64, None, None) == EINVAL
    mine = lib.mi_nerf_last_error()
    assert b"n_rays=-5" in mine
    seen = {}

    def other():
        lib.mi_nerf_stratified_z(4, 64, 2.0, 6.0, None, None, None)
        seen["msg"] = lib.mi_nerf_last_error()
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert b"t_rand" in seen["msg"] and lib.mi_nerf_last_error() == mine
