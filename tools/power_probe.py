"""Package power and shader clock while ONE kernel family runs back to back (rocm-smi sampled from a side thread): the evidence behind "the bf16 kernel is
power-limited" (DESIGN 3.3).  For each mode -- idle, fp32, f16s, bf16 fine-network launches of the 4096-ray batch -- ~4 s of launches, rocm-smi read every 0.4 s.
    python tools/power_probe.py  ->  profiles/r05_power_clock_by_kernel.txt"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights

dev = torch.device("cuda:0")
sd = synthetic.make_state_dict(0, 8, 256)
packed = weights.PackedNeRF.from_state_dict(sd, dev)
K, H, W = synthetic.lego_camera()
n, S = 4096, 192
pix = torch.from_numpy(synthetic.pixel_batch(H, W, n, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
raw = torch.empty(n, S, 4, device=dev)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        j = json.loads(r.stdout)
        c = next(iter(j.values()))
        pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k), float("nan"))
        sclk = next((v for k, v in c.items() if k.startswith("sclk")), "?")
        return pw, sclk
    except Exception as e:                                     # noqa: BLE001
        return float("nan"), repr(e)[:60]


def run(mode, seconds=4.0):
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.4)
    t = threading.Thread(target=sampler)
    t.start()
    t0, launches, ms = time.perf_counter(), 0, []
    while time.perf_counter() - t0 < seconds:
        if mode == "idle":
            time.sleep(0.2)
        else:
            blob = {"fp32": packed.fine, "bf16": packed.bf16()[1], "f16s": packed.f16s()[1]}[mode]
            it = {"fp32": 20, "bf16": 200, "f16s": 60}[mode]
            ms.append(ops.time_mlp_rays(packed.net, blob, rays, z, raw, it, mode == "bf16", 0, mode == "f16s"))
            launches += it
    stop.set(); t.join()
    pw = [p for p, _ in samples if p == p]
    clk = [c for _, c in samples]
    tail = ms[len(ms) // 2:] or [float("nan")]
    print(f"{mode:5s}: {launches:6d} launches, fine launch {sum(tail) / len(tail):.4f} ms (second half); package power {min(pw, default=float('nan')):.0f}-{max(pw, default=float('nan')):.0f} W "
          f"(median {sorted(pw)[len(pw) // 2] if pw else float('nan'):.0f}); sclk samples {clk[len(clk) // 2:][:6]}", flush=True)


for m in ("idle", "fp32", "f16s", "bf16", "idle"):
    run(m)


# the training step (fp32 and split precision): forward + backward + Adam on the same 4096-ray batch, back to back for ~5 s each
def run_train(f16s, seconds=5.0):
    from types import SimpleNamespace
    from nerf_pytorch_paeng_amd import nerf_process as NP
    from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder
    model = NeRF(8, 256, 63, 27).to(dev)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    posenc = get_positional_encoder(10), get_positional_encoder(4)
    opts = SimpleNamespace(near=2.0, far=6.0, N_samples_c=64, N_samples_f=128, perturb=1.0, chunk_rays=4096, chunk_pts=524288, data_type="blender", gpu_ids=[0], rank=0)
    target = torch.rand(n, 3, device=dev)
    optim = torch.optim.Adam(model.parameters(), lr=5e-4)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.4)
    t = threading.Thread(target=sampler)
    t.start()
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        rgb_c, _, rgb_f, _ = NP.batchify_rays_and_render_by_chunk(o, d, model, posenc, H, W, K, opts, f16s=f16s)
        optim.zero_grad()
        (torch.nn.functional.mse_loss(rgb_c, target) + torch.nn.functional.mse_loss(rgb_f, target)).backward()
        optim.step()
        steps += 1
        if steps % 20 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    stop.set(); t.join()
    pw = sorted(p for p, _ in samples if p == p)
    print(f"train{' f16s' if f16s else ' fp32'}: {steps} steps, {dt * 1e3:.2f} ms per step; package power median {pw[len(pw) // 2] if pw else float('nan'):.0f} W (max {pw[-1] if pw else float('nan'):.0f}); "
          f"sclk samples {[c for _, c in samples][len(samples) // 2:][:6]}", flush=True)


run_train(False)
run_train(True)
