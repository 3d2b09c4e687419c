"""Round 6 A/B: the bf16 step at small ray counts (auto launch plan, jitter drawn in the kernels), for two builds of the library in alternating processes on ONE box:
   MI_NERF_LIB=<the build before the fused middle> python tools/r06_fused_middle_probe.py ; python tools/r06_fused_middle_probe.py ; ...   (tools/r06_run3.sh)"""
import os,sys,time,statistics
sys.path.insert(0,'.')
import torch
from nerf_pytorch_paeng_amd import ops, synthetic, weights
dev=torch.device('cuda:0')
sd=synthetic.make_state_dict(0,8,256); packed=weights.PackedNeRF.from_state_dict(sd,dev)
K,H,W=synthetic.lego_camera(); pose=synthetic.pose_spherical(0.,-30.,4.)
blobs=packed.bf16()
for n in (256,512,1024,4096):
    pix=torch.from_numpy(synthetic.pixel_batch(H,W,n,0)).to(dev)
    o,d=ops.make_o_d_pixels(W,H,K,pose,pix); rays=torch.cat([o,d],-1).contiguous()
    res={}
    for name,ppw in (("auto",0),):
        cfg=ops.render_cfg(2.,6.,64,128,False,True,points_per_wave=ppw,seed=0)
        ws=torch.empty(ops.workspace_layout(cfg,n).total,dtype=torch.uint8,device=dev)
        out=(torch.empty(n,3,device=dev),torch.empty(n,device=dev),torch.empty(n,3,device=dev),torch.empty(n,device=dev))
        def step(): ops.render_rays(packed.net,blobs[0],blobs[1],cfg,rays,None,None,workspace=ws,out=out)
        t=time.perf_counter()
        while time.perf_counter()-t<0.1:
            for _ in range(8): step()
            torch.cuda.synchronize()
        ts=[]
        for r in range(5):
            t0=time.perf_counter()
            for _ in range(200): step()
            torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/200*1e6)
        res[name]=statistics.median(ts)
    print(os.environ.get('MI_NERF_LIB','shipped').split('/')[-1], n, {k:round(v,1) for k,v in res.items()}, flush=True)
