"""Multi-process (world_size 2, gloo, CPU) tests of the frame-parallel host logic: shard plans, the padded uneven
all-gather of frame slices, the GroupNorm partial-sum all-reduce.  The kernels themselves are covered by -m gpu tests
(tests/test_dist_gpu.py runs the sharded UNet with two ranks on one GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lkgd_amd.dist import allreduce_sums, gather_frames, make_plan, split_frames


def test_split_and_plans():
    assert split_frames(14, 4) == (4, 4, 3, 3) and split_frames(14, 2) == (7, 7) and split_frames(14, 1) == (14,)
    assert split_frames(25, 4) == (7, 6, 6, 6)
    with pytest.raises(ValueError):
        split_frames(3, 4)
    p = make_plan(8, 6, 14, cfg=True)          # CFG(2) x frames(4,4,3,3)
    assert (p.cfg_groups, p.frame_shards, p.cfg_index, p.shard_index) == (2, 4, 1, 2)
    assert p.f0 == 8 and p.f_local == 3 and p.f_max == 4
    assert p.frame_group_ranks() == [4, 5, 6, 7] and p.cfg_partner_ranks() == [2, 6]
    p = make_plan(2, 1, 14, cfg=True)          # pure CFG-parallel: no frame exchange at all
    assert p.frame_shards == 1 and p.cfg_index == 1 and p.f_local == 14
    p = make_plan(4, 3, 14, cfg=False)         # guidance off: all ranks share the frames
    assert p.cfg_groups == 1 and p.splits == (4, 4, 3, 3) and p.f0 == 11
    covered = []
    for r in range(8):
        q = make_plan(8, r, 14, True)
        covered.append((q.cfg_index, q.f0, q.f0 + q.f_local))
    assert sorted(covered) == [(0, 0, 4), (0, 4, 8), (0, 8, 11), (0, 11, 14), (1, 0, 4), (1, 4, 8), (1, 8, 11),
                               (1, 11, 14)]
    with pytest.raises(ValueError):
        make_plan(3, 0, 14, cfg=True)
    # round 6: frame slices cut at multiples of `frame_unit` (the FSM hook fuses frames 2k / 2k+1: a pair lives on one rank)
    assert split_frames(14, 4, 2) == (4, 4, 4, 2) and split_frames(14, 2, 2) == (8, 6) and split_frames(8, 4, 2) == (2, 2, 2, 2)
    assert split_frames(14, 7, 2) == (2,) * 7
    with pytest.raises(ValueError):
        split_frames(14, 8, 2)                 # more shards than pairs
    with pytest.raises(ValueError):
        split_frames(13, 2, 2)                 # an odd clip has no whole pairs
    # ... and symmetric slices for flip=True joint attention: shard i and k-1-i mirror each other frame for frame
    assert split_frames(14, 4, symmetric=True) == (4, 3, 3, 4) and split_frames(7, 3, symmetric=True) == (2, 3, 2)
    assert split_frames(6, 4, symmetric=True) == (2, 1, 1, 2) and split_frames(14, 2, symmetric=True) == (7, 7)
    with pytest.raises(ValueError):
        split_frames(5, 2, symmetric=True)
    p = make_plan(8, 7, 14, cfg=True, frame_unit=2)
    assert p.splits == (4, 4, 4, 2) and p.f0 == 12 and p.f_local == 2 and all(s % 2 == 0 for s in p.splits)
    # one frame slice (pure CFG-parallel) with several entries per rank: nothing to exchange, the tokens come back as they are
    from lkgd_amd.dist_run import ShardInfo
    sh = ShardInfo(make_plan(2, 1, 6, cfg=True), None, entries=2)
    x = torch.arange(2 * 6 * 4 * 3, dtype=torch.float32).reshape(-1, 3)
    assert sh.gather(x) is x and sh.to_pixels(x, 4) is x and sh.to_frames(x, 4) is x and (sh.b0, sh.B_total) == (2, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, frames, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(max(1, int(os.environ.get("LKGD_TEST_HOST_CPUS", os.cpu_count() or 8)) // world))   # the ranks share the box's host cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = make_plan(world, rank, frames, cfg=False)
        full = torch.arange(frames * 3 * 5, dtype=torch.float32).reshape(frames, 3, 5)
        local = full[plan.f0:plan.f0 + plan.f_local].clone()
        got = gather_frames(local, plan)
        ok = torch.equal(got, full)
        sums = torch.full((1, 32, 2), float(rank + 1))
        allreduce_sums(sums, plan)
        ok = ok and torch.equal(sums, torch.full((1, 32, 2), float(sum(range(1, world + 1)))))
        # fp16 payload, as the activations are
        h = gather_frames(local.half(), plan)
        ok = ok and torch.equal(h, full.half())
        # boundary-frame exchange for the Conv3d halo: slots 0 / f_local+1 <- neighbours' last / first frame, zero at the ends
        from lkgd_amd.dist import exchange_halo
        hb = torch.zeros(plan.f_local + 2, 3, 5)
        hb[1:plan.f_local + 1] = local
        exchange_halo(hb, plan)
        padded = torch.cat([torch.zeros(1, 3, 5), full, torch.zeros(1, 3, 5)])
        ok = ok and torch.equal(hb, padded[plan.f0:plan.f0 + plan.f_local + 2])
        # both forms of the exchange (default all-gather of boundary frames, opt-in neighbour-only P2P) give the same buffer,
        # also as recorded steps replayed on new values (P2P stays opt-in until compared on a multi-GPU RCCL world)
        import lkgd_amd.dist as ld
        from lkgd_amd import replay as _rp
        for allgather in (True, False):
            ld._HALO_ALLGATHER = allgather
            hb2 = torch.zeros(plan.f_local + 2, 3, 5)
            hb2[1:plan.f_local + 1] = local
            with _rp.record() as rec_h:
                exchange_halo(hb2, plan)
            ok = ok and torch.equal(hb2, hb)
            hb2[1:plan.f_local + 1] = 3.0 * local
            rec_h.run()
            want = 3.0 * padded[plan.f0:plan.f0 + plan.f_local + 2]
            ok = ok and torch.equal(hb2, want)
        ld._HALO_ALLGATHER = True
        # pixel re-sharding around the temporal attention (SURVEY.md 8e variant iii): [f_local, HW, C] -> [F, HW/k, C] and back
        from lkgd_amd.dist import frames_to_pixels, pixel_splits, pixels_to_frames
        for HW in (32, 7, 48):
            if HW < world:
                continue
            fullp = torch.arange(frames * HW * 3, dtype=torch.float32).reshape(frames, HW, 3) * 0.5 - 7.0
            mine = fullp[plan.f0:plan.f0 + plan.f_local].clone()
            px = pixel_splits(HW, world)
            ok = ok and sum(px) == HW and all(p > 0 for p in px)
            p0 = sum(px[:plan.shard_index])
            with _rp.record() as rec_p:
                xp = frames_to_pixels(mine, plan)
                back = pixels_to_frames(xp, plan, HW)
            ok = ok and torch.equal(xp, fullp[:, p0:p0 + px[plan.shard_index]])
            ok = ok and torch.equal(back, mine)
            mine.mul_(-2.0)                      # replay on new values in the same buffers
            rec_p.run()
            ok = ok and torch.equal(xp, -2.0 * fullp[:, p0:p0 + px[plan.shard_index]]) and torch.equal(back, mine)
        ok = ok and pixel_splits(9216, 4) == (2304, 2304, 2304, 2304) and pixel_splits(144, 4) == (48, 32, 32, 32)
        # recorded exchange steps (lkgd_amd/replay.py): new values in the same buffers, same plan
        from lkgd_amd import replay
        src = local.clone()
        sums2 = torch.full((1, 32, 2), float(rank + 1))
        with replay.record() as rec:
            got2 = gather_frames(src, plan)
            allreduce_sums(sums2, plan)
        ok = ok and torch.equal(got2, full)
        src.mul_(2.0)
        sums2.fill_(float(10 * (rank + 1)))
        rec.run()
        ok = ok and torch.equal(got2, full * 2.0)
        ok = ok and torch.equal(sums2, torch.full((1, 32, 2), float(10 * sum(range(1, world + 1)))))
        # several batch entries per rank (round 5: both clips of the [start, end] pair on every rank of a CFG half): the exchanges
        # run entry by entry on the (entry, frame, pixel) row layout of the token matrices
        from lkgd_amd.dist_run import ShardInfo
        sh = ShardInfo(plan, None, entries=2)
        ok = ok and sh.B_total == 2 and sh.b0 == 0
        HW, C = 6, 3
        if HW >= world:
            both = torch.arange(2 * frames * HW * C, dtype=torch.float32).reshape(2, frames, HW, C)
            mine2 = both[:, plan.f0:plan.f0 + plan.f_local].reshape(-1, C).contiguous()
            full2 = sh.gather(mine2)
            ok = ok and torch.equal(full2, both.reshape(-1, C))
            px = pixel_splits(HW, world)
            p0 = sum(px[:plan.shard_index])
            xp2 = sh.to_pixels(mine2, HW)
            ok = ok and torch.equal(xp2, both[:, :, p0:p0 + px[plan.shard_index]].reshape(-1, C))
            ok = ok and torch.equal(sh.to_frames(xp2, HW), mine2)
            hb3 = torch.zeros(2, plan.f_local + 2, HW, C)
            hb3[:, 1:plan.f_local + 1] = both[:, plan.f0:plan.f0 + plan.f_local]
            sh.halo(hb3.reshape(-1, C))
            pad2 = torch.cat([torch.zeros(2, 1, HW, C), both, torch.zeros(2, 1, HW, C)], dim=1)
            ok = ok and torch.equal(hb3, pad2[:, plan.f0:plan.f0 + plan.f_local + 2])
        # temporal GroupNorm + Conv3d halo in ONE collective: raw boundary frames and the partial sums in the same all-gather
        from lkgd_amd.dist import SUMS_SLOT, gather_boundary_frames_and_sums
        HWc, Cc = 4, 8
        xb = (torch.arange(2 * frames * HWc * Cc, dtype=torch.float32).reshape(2, frames, HWc, Cc) * 0.25 - 3.0).half()
        mine3 = xb[:, plan.f0:plan.f0 + plan.f_local].contiguous()
        sm = torch.arange(2 * 64, dtype=torch.float32).reshape(2, 32, 2) * (rank + 1) + 0.125
        got = gather_boundary_frames_and_sums([mine3[b, 0] for b in range(2)], [mine3[b, -1] for b in range(2)], sm, plan)
        n = HWc * Cc
        ok = ok and tuple(got.shape) == (world, 2, 2 * n + SUMS_SLOT)
        for r in range(world):
            pr = make_plan(world, r, frames, cfg=False)
            for b in range(2):
                ok = ok and torch.equal(got[r, b, :n], xb[b, pr.f0].reshape(-1))
                ok = ok and torch.equal(got[r, b, n:2 * n], xb[b, pr.f0 + pr.f_local - 1].reshape(-1))
                want = torch.arange(2 * 64, dtype=torch.float32).reshape(2, 64)[b] * (r + 1) + 0.125
                ok = ok and torch.equal(got[r, b, 2 * n:].view(torch.float32), want)     # fp32 bits survive the fp16 carrier
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("frames", [4, 5, 14, 3])
def test_uneven_frame_gather_world2(frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert res == [(0, True), (1, True)]


def test_frame_exchanges_world4():
    """14 frames over 4 shards (4, 4, 3, 3): interior ranks have a neighbour on both sides of the Conv3d halo"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 14, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(4))
    assert res == [(r, True) for r in range(4)]


def test_thread_world_collectives_match_their_definitions():
    """tests/thread_world.py (the in-process stand-in for torch.distributed the eight-rank GPU test runs on): sub-groups,
    all-gather, all-to-all with uneven rows, all-reduce"""
    from thread_world import ThreadWorld, run_ranks
    W = 8
    tw = ThreadWorld(W)
    rows = [1, 2, 3, 1]                                   # rows every rank of a group sends to peer 0..3

    def fn(r):
        groups = [tw.new_group([0, 1, 2, 3]), tw.new_group([4, 5, 6, 7])]
        g, si = groups[r // 4], r % 4
        t = torch.full((2,), float(r))
        tw.all_reduce(t, group=g)
        buf = torch.empty(4 * 2)
        tw.all_gather_into_tensor(buf, torch.full((2,), float(r)), group=g)
        inp = torch.arange(7 * 2, dtype=torch.float32).reshape(7, 2) + 100 * r
        out = torch.empty(4 * rows[si], 2)
        tw.all_to_all_single(out, inp, [rows[si]] * 4, rows, group=g)
        tw.barrier()
        return t, buf, out

    res = run_ranks(tw, fn)
    for r, (t, buf, out) in enumerate(res):
        base = 4 * (r // 4)
        assert t.tolist() == [float(sum(range(base, base + 4)))] * 2
        assert buf.tolist() == [float(base + i) for i in range(4) for _ in range(2)]
        si, start = r % 4, sum(rows[:r % 4])
        want = torch.cat([(torch.arange(14, dtype=torch.float32).reshape(7, 2) + 100 * (base + p))[start:start + rows[si]]
                          for p in range(4)])
        assert torch.equal(out, want)


def _worker_replay_entries(rank, world, port, frames, q):
    """with several batch entries per rank the joined result of the per-entry exchanges must be refreshed by
    the recorded steps themselves - record, change the inputs in place, replay, compare"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lkgd_amd import replay
        from lkgd_amd.dist import pixel_splits
        from lkgd_amd.dist_run import ShardInfo
        plan = make_plan(world, rank, frames, cfg=False)
        ok = True
        for entries in (1, 2, 3):
            sh = ShardInfo(plan, None, entries=entries)
            HW, C = 8, 3
            both = torch.arange(entries * frames * HW * C, dtype=torch.float32).reshape(entries, frames, HW, C) + 1.0
            mine = both[:, plan.f0:plan.f0 + plan.f_local].reshape(-1, C).clone()
            px = pixel_splits(HW, world)
            p0 = sum(px[:plan.shard_index])
            with replay.record() as rec:
                full = sh.gather(mine)
                xp = sh.to_pixels(mine, HW)
                back = sh.to_frames(xp, HW)
            for scale in (1.0, -3.0, 0.5):
                if scale != 1.0:
                    mine.copy_(scale * both[:, plan.f0:plan.f0 + plan.f_local].reshape(-1, C))
                    rec.run()
                ok = ok and torch.equal(full, scale * both.reshape(-1, C))
                ok = ok and torch.equal(xp, scale * both[:, :, p0:p0 + px[plan.shard_index]].reshape(-1, C))
                ok = ok and torch.equal(back, mine)
            rec.release()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _worker_mirror(rank, world, port, frames, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lkgd_amd import replay
        from lkgd_amd.dist import exchange_with_mirror
        plan = make_plan(world, rank, frames, cfg=False, symmetric=True)
        mp_ = make_plan(world, world - 1 - rank, frames, cfg=False, symmetric=True)
        ok = plan.f_local == mp_.f_local
        HW, W = 3, 4
        full = torch.arange(frames * HW * W, dtype=torch.float32).reshape(frames, HW, W) + 1.0
        mine = full[plan.f0:plan.f0 + plan.f_local].reshape(-1, W).clone()
        with replay.record() as rec:
            got = exchange_with_mirror(mine, plan)
        for scale in (1.0, -2.0):
            if scale != 1.0:
                mine.copy_(scale * full[plan.f0:plan.f0 + plan.f_local].reshape(-1, W))
                rec.run()
            ok = ok and torch.equal(got, scale * full[mp_.f0:mp_.f0 + mp_.f_local].reshape(-1, W))
            # local frame t of the mirror shard is global frame F-1-(f0 + f_local-1-t): the flip of patch/patch.py:471-475
            t = 0
            ok = ok and torch.equal(got.reshape(plan.f_local, HW, W)[plan.f_local - 1 - t], scale * full[frames - 1 - (plan.f0 + t)])
        bad = make_plan(world, rank, frames, cfg=False)            # (3, 3, 2, 2)-style slices are not mirror images of each other
        if bad.splits != plan.splits:
            try:
                exchange_with_mirror(mine, bad)
                ok = False
            except ValueError:
                pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames", [(4, 6), (3, 7), (2, 6)])
def test_mirror_exchange_of_symmetric_slices(world, frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mirror, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,frames", [(2, 5), (4, 6)])
def test_recorded_multi_entry_exchanges_follow_their_inputs(world, frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_replay_entries, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res == [(r, True) for r in range(world)]
