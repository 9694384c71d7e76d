import os, sys, torch, socket
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch.multiprocessing as mp
import torch.distributed as dist
import test_dist_gpu as T

def worker(rank, world, port, frames, steps, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lkgd_amd.dist_run import DistDenoiser
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline
    dev = torch.device("cuda", 0)
    pipe = StableVideoDiffusionPipeline(unet=T._build(dev))
    lat0, img, emb, ids = T._inputs(frames, True)
    pipe.scheduler.set_timesteps(steps)
    s0 = float(pipe.scheduler.init_noise_sigma)
    runner = DistDenoiser(pipe, world, rank, frames, cfg=True)
    out = runner.denoise((lat0*s0).half().to(dev), img.half().to(dev), emb.half().to(dev), ids.to(dev), steps, 1.0, 3.0)
    if rank == 0:
        ref = pipe.denoise((lat0*s0).half().to(dev), img.half().to(dev), emb.half().to(dev), ids.to(dev), steps, 1.0, 3.0)
        ref2 = pipe.denoise((lat0*s0).half().to(dev), img.half().to(dev), emb.half().to(dev), ids.to(dev), steps, 1.0, 3.0)
        # perturb the input by one fp16 ulp-ish to see the net's noise amplification
        lat1 = (lat0*s0).half().to(dev); lat1 = lat1 * (1 + 2**-11)
        ref3 = pipe.denoise(lat1.half(), img.half().to(dev), emb.half().to(dev), ids.to(dev), steps, 1.0, 3.0)
        r = lambda a,b: ((a.float()-b.float()).norm()/b.float().norm()).item()
        q.put((steps, r(out, ref), r(ref2, ref), r(ref3, ref)))
    dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    for steps in (1, 2, 4):
        q = ctx.Queue(); s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [ctx.Process(target=worker, args=(r, 4, port, 5, steps, q)) for r in range(4)]
        [p.start() for p in ps]
        print("steps, sharded-vs-single, rerun, ulp-perturbed:", q.get(timeout=300), flush=True)
        [p.join() for p in ps]
