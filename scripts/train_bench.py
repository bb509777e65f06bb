"""Training-step benchmark of the PSMNet cost path: features -> cat volume -> aggregator (BatchNorm in
training mode) -> fused up-sampling + soft-argmin -> weighted smooth-L1, backward through the HIP kernels, gradient
exchange (one flat buffer, one collective when launched with torch.distributed.run) and a torch optimizer step.
Synthetic features / ground truth; the reference trains PSMNet on 256 x 512 crops (configs/PSMNet/scene_flow.py).

    python scripts/train_bench.py [--batch 4] [--height 256] [--width 512] [--steps 5] [--warmup 2]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.dist_utils import FlatGradients, all_reduce_grads
from densematchingbenchmark_amd.modeling import build_model
from densematchingbenchmark_amd import synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", action="store_true", help="start from images: the HIP backbone trains too")
    ap.add_argument("--foreach-adam", action="store_true", help="torch's multi-tensor Adam (11 launches) instead of its fused one (1)")
    ap.add_argument("--config", default=os.path.join("PSMNet", "scene_flow.py"), help="relative to configs/ (PSMNet/scene_flow.py, "
                    "AcfNet/scene_flow_uniform.py, AcfNet/scene_flow_adaptive.py)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", args.config))
    model = build_model(cfg, backbone="hip" if args.images else None).to(dev)
    synthetic.init_params_(model, seed=0)
    model.train()
    flat = FlatGradients(model)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, fused=not args.foreach_adam)
    g = torch.Generator(device="cpu").manual_seed(1 + local)
    B, H, W = args.batch, args.height, args.width
    lf = torch.randn((B, 32, H // 4, W // 4), generator=g).to(dev)
    rf = torch.randn((B, 32, H // 4, W // 4), generator=g).to(dev)
    gt = (torch.rand((B, 1, H, W), generator=g) * 180.0 + 1.0).to(dev)
    batch = dict(leftFeature=lf, rightFeature=rf, leftDisp=gt)
    if args.images:
        batch = dict(leftImage=torch.randn((B, 3, H, W), generator=g).to(dev), rightImage=torch.randn((B, 3, H, W), generator=g).to(dev),
                     leftDisp=gt)

    def step():
        flat.zero_()
        _, losses = model(batch)
        loss = sum(losses.values())
        loss.backward()
        if world > 1:
            all_reduce_grads(model)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # phase split of one more step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    flat.zero_()
    ev[0].record()
    _, losses = model(batch)
    loss = sum(losses.values())
    ev[1].record()
    loss.backward()
    ev[2].record()
    if world > 1:
        all_reduce_grads(model)
    opt.step()
    ev[3].record()
    torch.cuda.synchronize()
    if local == 0:
        print(args.config + (" images -> loss" if args.images else " cost-path") + " training step: batch %d x %dx%d per GPU, %d GPU(s): %.1f ms/step = %.1f pairs/s; "
              "forward %.1f ms, backward %.1f ms, exchange+optimizer %.1f ms; loss %.4f; peak memory %.1f GB" %
              (B, H, W, world, ms, B * world / ms * 1e3, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]),
               ev[2].elapsed_time(ev[3]), float(loss.detach()), torch.cuda.max_memory_allocated(dev) / 2 ** 30), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
