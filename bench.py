#!/usr/bin/env python3
"""Headline benchmark: denoised frames/sec of the SVD Euler loop (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one clip of synthetic input: the full 25-step Euler sampling loop over the
real-shaped SVD UNet (random-init, fp16) for 14 frames x 576x1024 (latents 72x128), CFG on, inputs resident in HBM,
``output_type="latent"`` (CLIP / VAE are boundary stages and excluded).  value = frames of all K timed
clips / wall time (max over ranks).  With N > 1 the frames of the ONE clip are sharded over the ranks
(lkgd_amd/dist.py: CFG-parallel x frame slices, RCCL) - strong scaling.

Prints ONE JSON line on rank 0, with
  roofline     - the dominant kernel (the MFMA GEMM / implicit-conv kernel): algorithmic TFLOP/s measured live with
                 HIP events around every launch inside the timed region, against the dense fp16 MFMA peak;
  cpu_baseline - the fp32 CPU oracle (oracle/, "port") timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch

MFMA_PEAK_TFLOPS = 2500.0     # dense fp16/bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
EVENT_CLIP_STRIDE = 4         # GEMM launches are timed with HIP events on clips 0, 4, 8, ... of the timed region
UNET_TFLOP_C2 = 89.69         # algorithmic TFLOP of one UNet forward at C2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed clips")
    ap.add_argument("--warmup", type=int, default=1, help="untimed clips")
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--inference-steps", type=int, default=25)
    ap.add_argument("--tiny", action="store_true", help="tiny UNet config (plumbing checks only; not a valid number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-full", action="store_true",
                    help="cpu_baseline: do NOT time the full configs[1] forward of the fp32 oracle in this run (~4 minutes in a "
                         "CPU-only child process after the GPU work); quote the committed round-2 measurement beside a bounded sample")
    ap.add_argument("--cpu-full", action="store_true", help="(default since round 5; kept for old command lines)")
    ap.add_argument("--cpu-full-timeout", type=int, default=330, help="wall-clock guard of the full CPU forward, seconds")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed) VAE encode / decode measurement")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--lk", action="store_true",
                    help="configs[2]: the LKGD UNet (UNetSpatioTemporalConditionModel) with domain / flow features fused into the "
                         "CLIP embedding (not the headline workload)")
    ap.add_argument("--joint", action="store_true",
                    help="the `patch` joint-attention hooks (utils/util.py:561-606): TWO clips per call (the [start, end] pair of "
                         "the trans pipelines), masks [0,1,0,1], spatial + temporal joint branch (not the headline workload); with "
                         "--gpus N a rank holds its frame slice of both clips of its CFG half")
    ap.add_argument("--cogvideox", action="store_true",
                    help="configs[4]: the CogVideoX-2B image-to-video DiT loop (49 frames x 720x480 -> 13 latent frames x 60x90, CFG, "
                         "DDIM; NOT the headline workload, single GPU); a 'step' is one clip of --inference-steps DiT steps")
    ap.add_argument("--controlnet", action="store_true",
                    help="also run the ControlNet-SVD encoder every step (SURVEY.md 8f rank 1; not the headline workload)")
    return ap.parse_args()


def synthetic_inputs(dev, frames, h, w):
    """SURVEY.md 8d / BASELINE.md 2.1"""
    def rn(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    lat0 = rn((1, frames, 4, h, w), 12345)
    img = rn((1, 4, h, w), 12346) * 0.18215
    img = torch.cat([torch.zeros_like(img), img]).unsqueeze(1).repeat(1, frames, 1, 1, 1)
    emb = torch.cat([torch.zeros(1, 1, 1024), rn((1, 1, 1024), 12347)])
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    return lat0.to(dev), img.half().to(dev).contiguous(), emb.half().to(dev), ids.to(dev)


def build_unet(dev, tiny, lk=False):
    from lkgd_amd import unet as pu
    if tiny:
        cfg = pu.UNetConfig(sample_size=8, block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2),
                            addition_time_embed_dim=64, projection_class_embeddings_input_dim=192, num_frames=4)
    else:
        cfg = pu.UNetConfig()
    with torch.device("meta"):
        m = (pu.UNetSpatioTemporalConditionModel if lk else pu.UNetSpatioTemporalConditionControlNetModel)(cfg)
    m = m.to(torch.float16).to_empty(device=dev)
    pu.init_synthetic_weights_(m, seed=0)
    m.prepare()
    return m


def kernels_sha16():
    """fingerprint of the kernel sources (lkgd_amd/csrc/*.hip, *.h, *.inc): stamped into profiles/pmc_traffic.json when the
    PMC passes are summarised (tools/pmc_summary.py) and compared here, so a stale traffic figure says so"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(REPO, "lkgd_amd", "csrc", "*"))):
        if fn.endswith((".hip", ".h", ".inc")):
            with open(fn, "rb") as f:
                h.update(os.path.basename(fn).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_full_forward_child(args):
    """ONE full configs[1] forward of the fp32 oracle (89.69 TFLOP) on this box's host cores, in a fresh CPU-only child
    process (tools/cpu_full_forward.py: imports torch + oracle/, never touches the GPU), started AFTER the GPU work so that
    its threads do not take the launch thread's core inside the timed region.  Wall-clock guard: the child is killed at
    --cpu-full-timeout seconds and None is returned (the caller falls back to the bounded sample + the quoted measurement)."""
    import subprocess
    env = dict(os.environ, FRAMES=str(args.frames), LAT_H=str(args.height // 8), LAT_W=str(args.width // 8),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", LKGD_PROGRESS_STDERR="1")
    t0 = time.time()
    print(f"bench.py: timing ONE full configs[1] forward of the fp32 oracle on the host cores (CPU-only child process, guard "
          f"{args.cpu_full_timeout} s; --no-cpu-full skips it) ...", file=sys.stderr, flush=True)
    try:     # the child's per-block progress lines go to this process's stderr: a multi-minute leg that is never silent
        p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "cpu_full_forward.py")], env=env, stdout=subprocess.PIPE,
                           stderr=None, timeout=args.cpu_full_timeout, text=True)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the full CPU forward did not finish within {args.cpu_full_timeout} s - falling back", file=sys.stderr)
        return None
    if p.returncode != 0:
        print(f"bench.py: tools/cpu_full_forward.py exited with {p.returncode} - falling back", file=sys.stderr)
        return None
    for ln in reversed(p.stdout.strip().splitlines()):
        if ln.startswith("{"):
            rec = json.loads(ln)
            rec["child_wall_s"] = round(time.time() - t0, 1)
            return rec
    return None


def cpu_baseline(args):
    """the fp32 CPU oracle (oracle/, "port") timed on this box's host cores.  Default: ONE full configs[1] UNet forward (the
    whole workload of one Euler step: ~190 s on the GPU boxes seen so far) measured by THIS run in a CPU-only child process,
    x 25 steps / 14 frames.  Fallback (--no-cpu-full, or the guard fired): a bounded 1.19-TFLOP sample measured here beside
    the full-forward figure quoted from profiles/r02_cpu_full_forward.json, and the line says which it is."""
    from tools.hostcpus import cpu_facts, host_cpus
    cores = host_cpus()              # what this process can really run (cgroup quota / affinity), not os.cpu_count()
    torch.set_num_threads(cores)
    if args.tiny:
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "port", "sample": "tiny config (invalid)"}
    if not args.no_cpu_full:
        full = cpu_full_forward_child(args)
        if full and full.get("seconds"):
            fps_full = args.frames / (args.inference_steps * float(full["seconds"]))
            return {"value": round(fps_full, 6),
                    "unit": "frames/s (one full configs[1] UNet forward of the fp32 oracle, measured, x 25 steps / 14 frames)",
                    "cores": full.get("threads", cores), "kind": "port", "value_source": "measured by this run",
                    "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": full.get("cgroup_cpu_quota"),
                    "sample": f"ONE full configs[1] forward (CFG 2 x {args.frames} frames x {args.height // 8}x{args.width // 8} latent, "
                              f"{UNET_TFLOP_C2:.2f} TFLOP) of the fp32 oracle (torch eager) in {float(full['seconds']):.1f} s = "
                              f"{full.get('tflops')} TFLOP/s on {full.get('threads', cores)} host threads, in a CPU-only child process "
                              f"after the GPU work ({full.get('child_wall_s')} s with the model build)",
                    "finite": full.get("finite")}
    from oracle import unet as ou
    t0 = time.time()
    with torch.device("meta"):
        o = ou.UNetSpatioTemporalConditionControlNetModel(ou.SVD_CONFIG)
    o = o.to_empty(device="cpu")
    with torch.no_grad():
        for p in o.parameters():
            if p.ndim >= 2:
                p.normal_(0.0, 1.0 / p[0].numel() ** 0.5)
            else:
                p.fill_(0.0)
        for m in o.modules():
            if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)):
                m.weight.fill_(1.0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 2, 8, 32, 32, generator=g)
    enc = torch.randn(2, 1, 1024, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    t_build = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        o(x, torch.tensor(1.0), enc, added_time_ids=ids, return_dict=False)
    dt = time.time() - t0
    tflop_sample = 2.37 / 2
    tflops = tflop_sample / dt
    fps = args.frames / (args.inference_steps * UNET_TFLOP_C2 / tflops)
    full, full_src = None, None
    try:        # the full forward timed on a GPU box's host in round 2 (tools/cpu_full_forward.py, committed log)
        with open(os.path.join(REPO, "profiles", "r02_cpu_full_forward.json")) as f:
            full = json.load(f)
        full_src = "NOT measured by this run: quoted from profiles/r02_cpu_full_forward.json (tools/cpu_full_forward.py " \
                   "on a GPU box's host, round 2); this run measured `live_sample` only" + (
                       " (--no-cpu-full)" if args.no_cpu_full else " - its full forward hit the wall-clock guard")
    except (OSError, ValueError):
        pass
    live = (f"oracle fp32 (torch eager), one UNet forward of the real-shaped SVD UNet at CFG-batch 2 x 2 frames x 32x32 latent "
            f"(half of config 1's geometry, {tflop_sample:.2f} TFLOP) in {dt:.1f} s = {tflops:.3f} TFLOP/s on {cores} threads "
            f"(os.cpu_count={os.cpu_count()}) -> {fps:.6f} C2-equivalent frames/s by FLOPs; model build {t_build:.0f} s not counted")
    if full and full.get("seconds"):
        # the small sample under-feeds the host's threads and extrapolates 3x low: the quoted full forward is the stated value
        fps_full = args.frames / (args.inference_steps * float(full["seconds"]))
        return {"value": round(fps_full, 6), "unit": "frames/s (one full configs[1] UNet forward of the fp32 oracle, measured, x 25 steps / 14 frames)",
                "cores": full.get("threads", cores), "kind": "port", "value_source": full_src, "cpu_model": _cpu_model(),
                "sample": f"ONE full configs[1] forward (CFG 2 x 14 frames x 72x128 latent, {UNET_TFLOP_C2:.2f} TFLOP) of the fp32 oracle "
                          f"in {float(full['seconds']):.1f} s on {full.get('threads', cores)} host threads",
                "live_sample": live, "live_sample_value": round(fps, 6)}
    return {"value": round(fps, 6), "unit": "frames/s (C2-equivalent, extrapolated by algorithmic FLOPs)",
            "cores": cores, "kind": "port", "value_source": "measured by this run (bounded sample)", "cpu_model": _cpu_model(),
            "sample": live}


def vae_stages(dev, args, latents, loop_s_per_clip):
    """outside the timed region: the clip-level stages either side of the loop on the HIP path (lkgd_amd/vae.py, random-init
    SVD VAE shapes) - encode of the conditioning image, temporal decode of the clip's denoised latents in one chunk - and
    the end-to-end videos/s they imply next to the loop-only metric"""
    from lkgd_amd import unet as pu
    from lkgd_amd import vae as pv
    with torch.device("meta"):
        v = pv.AutoencoderKLTemporalDecoder()
    v = v.to(torch.float16).to_empty(device=dev)
    pu.init_synthetic_weights_(v, seed=2)
    img = torch.rand(1, 3, args.height, args.width, generator=torch.Generator().manual_seed(5)).to(dev) * 2 - 1
    z = (latents[0].float() / v.config.scaling_factor).clamp(-30, 30).half()          # [F, 4, h, w]

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r
    t_enc, _ = timed(lambda: v.encode(img).latent_dist.mode())
    t_dec, frames = timed(lambda: v.decode(z, num_frames=args.frames).sample)
    # the CLIP image encoder of `image_encoder/` (ViT-H/14, random init) on the HIP path: lkgd_amd/clip.py
    from lkgd_amd import clip as pc
    with torch.device("meta"):
        enc = pc.CLIPVisionModelWithProjection()
    enc = enc.to(torch.float16).to_empty(device=dev)
    pu.init_synthetic_weights_(enc, seed=3)
    with torch.no_grad():
        enc.vision_model.embeddings.position_embedding.weight.normal_(0, 0.02)
        enc.vision_model.embeddings.class_embedding.normal_(0, 0.02)
    pix = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(7)).to(dev)
    t_clip, emb = timed(lambda: enc(pix).image_embeds)
    total = loop_s_per_clip + t_enc + t_dec + t_clip
    return {"clip_encode_ms": round(t_clip * 1e3, 2), "clip_finite": bool(torch.isfinite(emb.float()).all().item()),
            "vae_encode_ms": round(t_enc * 1e3, 2), "vae_decode_ms": round(t_dec * 1e3, 2),
            "decode_chunk_size": args.frames, "decoded": list(frames.shape),
            "finite": bool(torch.isfinite(frames.float()).all().item()),
            "videos_per_s": round(1.0 / total, 4), "frames_per_s": round(args.frames / total, 4),
            "note": "CLIP-ViT-H image embedding + VAE encode + loop + temporal VAE decode of one clip, all on the HIP path; "
                    "untimed extras, the headline value is the loop alone"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent never touches
    the GPU - children are created before any HIP call), rank 0's JSON line goes to the inherited stdout; exit code =
    the worst child's"""
    import socket
    import subprocess
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:          # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    return rc


def bench_cogvideox(args):
    """configs[4] on one GPU (lkgd_amd/cogvideox.py): random-init 1.69 B-parameter DiT + LK modules, synthetic latents / prompt
    embeddings; prints its own JSON line (frames of the decoded video per second: 49 per clip)"""
    from lkgd_amd import cogvideox as pc
    from lkgd_amd import unet as pu
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("LKGD_DIST_BACKEND", "nccl"))
    cfg = pc.DiTConfig(in_channels=32) if not args.tiny else pc.DiTConfig(
        num_attention_heads=2, in_channels=32, time_embed_dim=64, num_layers=2, sample_width=12, sample_height=8, sample_frames=9,
        max_text_seq_length=16)
    with torch.device("meta"):
        m = pc.CogVideoXTransformer3DModel(cfg)
    m = m.to(torch.float16).to_empty(device=dev)
    pu.init_synthetic_weights_(m, seed=0)
    with torch.no_grad():
        for n, p in m.named_parameters():      # adaLN modulation / gates small, as in a trained model's range
            if ".linear." in n and ("norm1" in n or "norm2" in n or "norm_out" in n):
                p.mul_(0.1)
    g = torch.Generator().manual_seed(3)
    f = (cfg.sample_frames - 1) // cfg.temporal_compression_ratio + 1
    lat = torch.randn(1, f, 16, cfg.sample_height, cfg.sample_width, generator=g).half().to(dev)
    img = (0.5 * torch.randn(1, f, 16, cfg.sample_height, cfg.sample_width, generator=g)).half().to(dev)
    pe = torch.randn(2, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g).half().to(dev)
    dom, flow = torch.randn(1, 1, 1000, generator=g).to(dev), torch.randn(1, 1, 1000, generator=g).to(dev)
    steps = args.inference_steps if args.inference_steps != 25 else 50
    sch = pc.CogVideoXDDIMScheduler()

    if world > 1:
        from lkgd_amd.dist_run import DistDiTDenoiser
        runner = DistDiTDenoiser(m, sch, world, rank, f)

        def one_clip():
            return runner.denoise(lat, img, pe, dom, flow, steps, 6.0, True)
    else:
        def one_clip():
            return pc.denoise(m, sch, lat, img, pe, dom, flow, steps, 6.0, True)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier(device_ids=[dev.index]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_clip()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_clip()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    L = cfg.max_text_seq_length + f * (cfg.sample_height // 2) * (cfg.sample_width // 2)
    D = cfg.num_attention_heads * 64
    tflop = 2 * cfg.num_layers * (2.0 * L * D * D * 12 + 4.0 * L * L * D) / 1e12        # CFG batch 2: projections + FF, attention
    line = {"metric": "decoded-video frames/sec of the CogVideoX-2B DiT loop (49f x 720x480, DDIM, CFG) - configs[4], NOT the headline",
            "value": round(args.steps * cfg.sample_frames / dt, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"CogVideoX-2B image-to-video DiT with the LK fuse, {cfg.sample_frames} frames x 720x480 "
                                   f"({f} latent frames x {cfg.sample_height}x{cfg.sample_width}, {L} joint tokens), {steps} DDIM steps, "
                                   "dynamic CFG 6.0" + (" [TINY - INVALID]" if args.tiny else ""),
                       "parallelism": "single GPU" if world == 1 else f"cfg x latent-frame shards over {world} GPUs"},
            "finite_output": bool(torch.isfinite(out.float()).all().item()),
            "dit_ms_per_forward": round(dt / args.steps / steps * 1e3, 2),
            "dit_tflops": round(tflop / (dt / args.steps / steps), 1), "roofline": None, "cpu_baseline": None}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.cogvideox:
        return bench_cogvideox(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    # LKGD_FORCE_DIST=1: take the sharded runner + RCCL path even with one rank (functional check on a 1-GPU box)
    distributed = world > 1 or os.environ.get("LKGD_FORCE_DIST", "0") == "1"
    comm_ranks = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # "nccl" == RCCL on ROCm.  LKGD_DIST_BACKEND=gloo is a functional-test knob for boxes with fewer GPUs than ranks
        backend = os.environ.get("LKGD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl")
        else:
            dist.init_process_group(backend)
        comm_ranks = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}   # read back from the communicator

    from lkgd_amd import ops
    from lkgd_amd.pipeline import StableVideoDiffusionPipeline

    h, w = args.height // 8, args.width // 8
    unet = build_unet(dev, args.tiny, args.lk)
    pipe = StableVideoDiffusionPipeline(unet=unet)
    dom = flow = None
    if args.lk:
        dom = torch.randn(1, 1, 1000, generator=torch.Generator().manual_seed(12348)).half().to(dev)
        flow = torch.randn(1, 1, 1000, generator=torch.Generator().manual_seed(12349)).half().to(dev)
    lat0, img, emb, ids = synthetic_inputs(dev, args.frames, h, w)
    nclips = 1
    if args.joint:
        from lkgd_amd import patch
        patch.apply_patch(pipe, with_temporal_block=True)
        patch.initialize_joint_layers(pipe)
        with torch.no_grad():                      # zero-init joint layers would be an identity branch: give them weights
            g = torch.Generator().manual_seed(12350)
            for name, prm in unet.named_parameters():
                if "attn1n" in name or "conv1n" in name:
                    prm.copy_((torch.randn(prm.shape, generator=g) * (0.5 / max(prm.shape[-1], 1) ** 0.5)).to(prm))
        unet.invalidate()
        patch.set_joint_attention_mask(pipe, [0, 1, 0, 1])
        nclips = 2
        lat0 = torch.cat([lat0, 0.9 * lat0.flip(1)])
        img = torch.stack([img[0], img[0], img[1], 0.8 * img[1]])        # [u1, u2, c1, c2]
        emb = torch.stack([emb[0], emb[0], emb[1], 0.8 * emb[1]])
        ids = ids[:1].repeat(4, 1)
        if dom is not None:                        # one feature row per batch entry [u1, u2, c1, c2]
            dom, flow = torch.cat([dom, 0.7 * dom] * 2), torch.cat([flow, 0.6 * flow] * 2)
    ctrl_cond = None
    if args.controlnet:
        from lkgd_amd import controlnet as pc
        from lkgd_amd import unet as pu
        with torch.device("meta"):
            cn = pc.ControlNetSDVModel(pu.UNetConfig(**{k: v for k, v in unet.config.__dict__.items()
                                                        if k in pu.UNetConfig.__dataclass_fields__}))
        cn = cn.to(torch.float16).to_empty(device=dev)
        pu.init_synthetic_weights_(cn, seed=1)
        pipe.controlnet = cn
        ctrl_cond = (2.0 * torch.rand(1, args.frames, 3, args.height, args.width,
                                      generator=torch.Generator().manual_seed(12348)) - 1.0).half().to(dev).repeat(2, 1, 1, 1, 1)
    pipe.scheduler.set_timesteps(args.inference_steps)
    sigma0 = float(pipe.scheduler.init_noise_sigma)

    if distributed:
        from lkgd_amd.dist_run import DistDenoiser
        runner = DistDenoiser(pipe, world, rank, args.frames)

        def one_clip():
            return runner.denoise((lat0 * sigma0).half(), img, emb, ids, args.inference_steps, 1.0, 3.0,
                                  domain_features=dom, flow_features=flow, controlnet_condition=ctrl_cond)
    else:
        def one_clip():
            return pipe.denoise((lat0 * sigma0).half(), img, emb, ids, args.inference_steps, 1.0, 3.0,
                                domain_features=dom, flow_features=flow, controlnet_condition=ctrl_cond)

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier(device_ids=[dev.index]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_clip()
    # per-launch HIP events of the GEMM family are recorded on every EVENT_CLIP_STRIDE-th timed clip (the first one
    # included): an event pair per launch is ~15 000 extra queue packets per clip, which the other clips are spared
    events = [] if not args.no_kernel_events else None
    clips_sampled = 0
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for c in range(args.steps):
        sample = events is not None and c % EVENT_CLIP_STRIDE == 0
        ops.GEMM_EVENTS = events if sample else None
        clips_sampled += int(sample)
        out = one_clip()
    ops.GEMM_EVENTS = None
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(out.float()).all().item())
    # working set of the timed region: live tensors at the peak (weights, packed weights, the forward's
    # activations, bench inputs), what the caching allocator holds for them, and the recorded forward's private pool alone
    owner = runner if distributed else pipe
    memory = {"peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
              "peak_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 1e9, 2),
              "plan_arena_gb": round(owner._arenas.reserved_bytes() / 1e9, 2)}

    roofline = None
    if events:
        tot_ms = sum(s.elapsed_time(e) for s, e, _ in events)
        tot_flop = sum(f for _, _, f in events)
        achieved = tot_flop / (tot_ms * 1e-3) / 1e12
        traffic = traffic_src = None      # HBM bytes per launch from the committed PMC passes (bench.py cannot collect PMC itself)
        try:
            with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            traffic = pj["gemm_family_bytes_per_launch"]
            here = kernels_sha16()
            traffic_src = {"file": "profiles/pmc_traffic.json", "kernels_sha16_when_collected": pj.get("kernels_sha16"),
                           "kernels_sha16_now": here, "same_kernel_sources": pj.get("kernels_sha16") == here}
            if pj.get("kernels_sha16") != here:
                print("bench.py: roofline.traffic comes from PMC passes over OTHER kernel sources (profiles/pmc_traffic.json: "
                      f"{pj.get('kernels_sha16')}, this tree: {here}) - rerun tools/prof_pmc_traffic.sh", file=sys.stderr)
        except (OSError, KeyError, ValueError):
            pass
        # sharded runs time the GEMM launches of every 5th Euler step only (DistDenoiser.event_stride): scale the share
        sampled = len([i for i in range(args.inference_steps) if i % runner.event_stride == runner.event_stride // 2]) \
            if distributed else args.inference_steps
        scale = args.inference_steps / max(sampled, 1)
        roofline = {"bound": "mfma", "kernel": "lkgd_gemm_{wide,resw,rowpanel,stream}_kernel + lkgd_gemm_kernel + ff_fused_kernel + tattn_block_kernel + ln_qkv_kernel (MFMA GEMM / implicit-conv family; the fused kernels count the flop of their matrix products)",
                    "achieved": round(achieved, 2),
                    "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "launches": len(events),
                    "avg_launch_us": round(tot_ms * 1e3 / len(events), 2),
                    "gemm_share_of_wall": round(tot_ms * 1e-3 * scale * args.steps / max(clips_sampled, 1) / dt, 3),
                    "launches_timed": f"every GEMM launch of {clips_sampled} of the {args.steps} timed clips" + (
                        "" if not distributed else f", {sampled} of {args.inference_steps} Euler steps each")}

    e2e = None
    if rank == 0 and world == 1 and not args.tiny and not args.no_vae:
        e2e = vae_stages(dev, args, out, dt / args.steps / nclips)
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
        fps = args.steps * nclips * args.frames / dt
        line = {
            "metric": "denoised frames/sec (14f x 576x1024, 25-step Euler)", "value": round(fps, 4),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"SVD {args.frames} frames x {args.height}x{args.width}, "
                                   f"{args.inference_steps} Euler steps, CFG 1.0->3.0, single clip, "
                                   + ("ControlNet pipeline loop (pipeline_stable_video_diffusion_controlnet, NOT the headline)"
                                      if args.controlnet else
                                      "LKGD UNet with domain / flow features (configs[2], NOT the headline)" if args.lk else
                                      "TWO clips with the patch joint-attention hooks, masks [0,1,0,1] (NOT the headline)" if args.joint else
                                      "vanilla pipeline_stable_video_diffusion_trans loop (configs[1])")
                                   + (" [TINY UNET - INVALID]" if args.tiny else ""),
                       "unet": "random-init SVD shapes (320,640,1280,1280), heads (5,10,20,20), 1.52 B params",
                       "parallelism": "single GPU" if world == 1 else f"cfg x frame shards over {world} GPUs"},
            "finite_output": finite, "rccl_ranks": comm_ranks, "memory": memory, "end_to_end": e2e,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
