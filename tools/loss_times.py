"""GPU time of the fused L1 + SSIM loss (tgs_l1_ssim: k_ssim_stats_stream + k_loss_reduce, then k_ssim_grad_stream) at the trainers' image sizes:
python tools/loss_times.py   -> us per call of the value pass alone and of value + gradient (HIP events around 50 calls each), and the check against
the torch restatement of loss_utils.py:39-63 on the smaller image."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from youreditableavatar_amd.loss import l1_ssim_value_and_grad
dev = torch.device("cuda", 0)
for (C, H, W) in ((3, 2048, 2048), (3, 1080, 1920)):
    torch.manual_seed(1)
    img, gt = torch.rand(C, H, W, device=dev), torch.rand(C, H, W, device=dev)
    res = {}
    for need in (False, True):
        for _ in range(5):
            l1_ssim_value_and_grad(img, gt, 0.2, need_grad=need)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            out3, grad = l1_ssim_value_and_grad(img, gt, 0.2, need_grad=need)
        e1.record(); torch.cuda.synchronize()
        res[need] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{C}x{H}x{W}: value {res[False]:.1f} us, value + gradient {res[True]:.1f} us (gradient pass {res[True] - res[False]:.1f} us), loss {out3[0].item():.6f}, |grad| {grad.abs().sum().item():.6f}")
