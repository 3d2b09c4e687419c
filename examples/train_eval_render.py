"""The reference's main loop (main.py:17-161: train -> test -> render) on this package, end to end on one MI355X, with a synthetic scene
standing in for the dataset loaders (no dataset ships with either repository).  Everything below `from nerf_pytorch_paeng_amd ...` keeps the
reference's names and call shapes (train.py:12, test.py:17, test.py:111); swap the imports back and the same script drives the reference.

    python examples/train_eval_render.py [--steps 2000] [--size 64] [--out /tmp/nerf_demo] [--precision fp32|f16s] [--net-width 256]
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pytorch_paeng_amd import harness, synthetic                                          # noqa: E402
from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder                          # noqa: E402  (model/NeRF.py, model/PositionalEncoding.py)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--size", type=int, default=64, help="training / test image side")
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--out", default="/tmp/nerf_demo")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16s"])
    ap.add_argument("--net-width", type=int, default=256)
    ap.add_argument("--render-views", type=int, default=8)
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H = W = a.size
    # config.py:105-111 -- the fields the path reads, with lego.txt's values
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288, data_type="blender",
                           gpu_ids=[0], rank=0, exp_name="demo", N_rays=1024, global_batch=True, idx_save=a.steps, idx_print=500, n_angle=a.render_views,
                           single_angle=-1, phi=-30.0, nf=4.0, precision=a.precision)
    K800, _, _ = synthetic.lego_camera()
    K = np.array([[K800[0][0] * W / 800.0, 0, W / 2], [0, K800[1][1] * H / 800.0, H / 2], [0, 0, 1]])
    posenc = get_positional_encoder(10), get_positional_encoder(4)                               # main.py:133
    poses = harness.get_render_pose(n_angle=a.views + 2, phi=-30.0, nf=4.0)                      # stand-in for load_blender's camera list
    # the "dataset": views of a fixed random NeRF rendered by the inference kernels (the teacher); the last two views are the test set
    teacher = NeRF(8, 256, 63, 27).to(dev)
    teacher.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic.make_state_dict(77, 8, 256).items()})
    with torch.no_grad():
        images = torch.stack([harness._render_pose(teacher, posenc, K, poses[i].to(dev), (H, W), opts)[0].reshape(H, W, 3) for i in range(a.views + 2)], 0)
    i_train, i_test = list(range(a.views)), [a.views, a.views + 1]

    model = NeRF(8, a.net_width, 63, 27, skips=[4]).to(dev)                                      # main.py:67-73
    optimizer = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))                # main.py:79-80
    criterion = torch.nn.MSELoss()
    getter = harness.global_batch(images, K, poses, i_train, (H, W), dev)                        # main.py:92-106
    t0 = time.perf_counter()
    for i in range(1, a.steps + 1):                                                              # main.py:136-139
        out = harness.train(i, i_train, images, (K, poses.numpy()), (H, W), model, criterion, posenc, optimizer, getter, None, opts, log_dir=a.out)
        if i % opts.idx_print == 0:
            torch.cuda.synchronize()
            print(f"step {i:6d}  loss {float(out['loss']):.5f}  psnr_f {float(out['psnr_f']):.2f} dB  {(time.perf_counter() - t0) / i * 1e3:.1f} ms/step", flush=True)
    fresh = NeRF(8, a.net_width, 63, 27, skips=[4]).to(dev)                                      # test() loads the checkpoint train() saved (test.py:20-21)
    res = harness.test(a.steps, i_test, posenc, fresh, images[i_test], K, poses[i_test].to(dev), (H, W), opts, log_dir=a.out,
                       save_dir=os.path.join(a.out, "test_result"))                              # main.py:140-149
    print(f"test: PSNR {['%.2f' % p for p in res['psnr']]} dB (mean {res['mean_psnr']:.2f}); PNGs and _result.txt in {a.out}/test_result")
    rgbs, disps = harness.render(a.steps, posenc, fresh, K, None, (H, W), opts, log_dir=a.out, save_dir=os.path.join(a.out, "render_result"))   # main.py:150-158
    print(f"render: {rgbs.shape[0]} frames {rgbs.shape[1]}x{rgbs.shape[2]} in {a.out}/render_result")
    return res


if __name__ == "__main__":
    main()
