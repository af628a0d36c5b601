"""Data-parallel host logic on CPU with the gloo backend, world_size 2 (the N>1 path of
bench.py / the trainer): bucketed gradient all-reduce (both the bucket-copy path and the
flat gradient-arena path), parameter broadcast, rank-sharded sampling, eval counters."""

from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepfakedetection_amd.arena import GradArena
from deepfakedetection_amd.dp import GradAllReducer, ShardedSampler, all_reduce_counts, broadcast_module_state


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # different init per rank on purpose
        model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
        broadcast_module_state(model)
        flat = torch.cat([t.detach().flatten().float() for t in list(model.parameters()) + list(model.buffers())])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "broadcast_module_state"

        params = list(model.parameters())
        # path 1: ordinary .grad tensors, tiny buckets so several collectives run
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        red = GradAllReducer(params, bucket_bytes=64)
        assert len(red.buckets) > 1 and red.grad_scale == 0.5
        red.reduce()
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, 3.0 * (i + 1))), "bucket path"
        # path 2: gradients living in the arena's flat buffer
        arena = GradArena(params)
        for i, (p, slot) in enumerate(zip(arena.params, arena.slots)):
            slot.fill_(float(rank + 1) * (i + 1))
            p.grad = slot.view_as(p)
        assert arena.holds_all_grads()
        GradAllReducer(params, bucket_bytes=48, arena=arena).reduce()
        for i, p in enumerate(arena.params):
            assert torch.allclose(p.grad, torch.full_like(p, 3.0 * (i + 1))), "arena path"
        arena.release()

        # path 3: hook-driven overlap — buckets are launched from inside backward() as their last
        # gradient appears; the result must equal the sum of the two ranks' local gradients
        torch.manual_seed(7 + rank)
        x = torch.randn(6, 7)
        model.zero_grad(set_to_none=True)
        model(x).square().sum().backward()
        local = [p.grad.clone() for p in params]
        gathered_g = []
        for g in local:
            parts = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            gathered_g.append(parts[0] + parts[1])
        hooked = GradAllReducer(params, bucket_bytes=64)
        hooked.attach()
        for _ in range(2):                                   # twice: the counters must re-arm
            model.zero_grad(set_to_none=True)
            model(x).square().sum().mul(0.5).backward()          # micro-batch 1 of 2: hooks stay quiet
            hooked.arm()
            model(x).square().sum().mul(0.5).backward()          # micro-batch 2 completes the step
            hooked.finish()
            for p, want in zip(params, gathered_g):
                assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), "hook path"
        assert hooked.launched_early >= len(hooked.buckets), "no bucket was launched from a hook"
        hooked.detach()

        total = all_reduce_counts(float(rank + 1), 10.0)
        assert total == [3.0, 20.0]
        sampler = ShardedSampler(11, rank, world, shuffle=True, seed=4)
        sampler.set_epoch(2)
        torch.save(list(iter(sampler)), os.path.join(out_dir, f"idx{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "idx0.pt"), torch.load(tmp_path / "idx1.pt")
    assert len(a) == len(b) == 6                                   # ceil(11/2), padded
    assert set(a) | set(b) == set(range(11))                       # every sample seen
    assert len(set(a) & set(b)) == 1                               # exactly the one padded duplicate


def _mixed_worker(rank: int, world: int, port: int) -> None:
    """the ranks decide on their own whether a cycle runs eagerly (hooks launch the buckets from inside
    backward) or replayed (one-shot reduce() afterwards) — a capture failure on ONE GPU must not make the ranks post different
    collectives.  Rank 0 takes the hook path, rank 1 the one-shot path, gradients in arena slots as on the GPU hot loop; several
    small buckets; the order in which autograd completes the buckets (last layer first) differs from the one-shot order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        model = torch.nn.Sequential(torch.nn.Linear(9, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4))
        params = list(model.parameters())
        arena = GradArena(params)
        red = GradAllReducer(params, bucket_bytes=160, arena=arena)
        assert len(red.buckets) >= 3 and red._ranges is not None
        red.attach()
        torch.manual_seed(50 + rank)
        x = torch.randn(5, 9)

        def backward_into_slots():
            for p in params:
                p.grad = None
            arena.reset()
            loss = model(x).square().sum()
            grads = torch.autograd.grad(loss, params)
            for p, slot, g in zip(arena.params, arena.slots, grads):
                slot.copy_(g.reshape(-1))
            return [g.clone() for g in grads]

        local = backward_into_slots()
        want = []
        for g in local:
            parts = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(parts, g)
            want.append(parts[0] + parts[1])
        for cycle in range(3):
            backward_into_slots()
            if (rank + cycle) % 2 == 0:
                # eager cycle: the hooks fire as each parameter's gradient lands in its slot, late layers first
                red.arm()
                for p, slot in reversed(list(zip(arena.params, arena.slots))):
                    p.grad = slot.view_as(p)
                    red._on_grad(p)
                red.finish()
            else:
                # replayed cycle: every gradient already sits in the arena, one call afterwards
                for p, slot in zip(arena.params, arena.slots):
                    p.grad = slot.view_as(p)
                red.finish()
            for p, w in zip(params, want):
                assert torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6), f"cycle {cycle}: mixed eager / replay ranks disagree"
        assert red.launched_early >= 1
        red.detach()
        arena.release()
    finally:
        dist.destroy_process_group()


def test_ranks_in_different_modes_post_the_same_collectives():
    mp.spawn(_mixed_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_sharded_sampler_properties():
    s0 = ShardedSampler(10, 0, 4, shuffle=False)
    assert list(s0) == [0, 4, 8] and len(s0) == 3
    assert list(ShardedSampler(10, 3, 4, shuffle=False)) == [3, 7, 1]          # wraps to pad
    d = ShardedSampler(10, 1, 4, shuffle=False, drop_last=True)
    assert list(d) == [1, 5] and len(d) == 2
    e1, e2 = ShardedSampler(50, 0, 2, seed=1), ShardedSampler(50, 0, 2, seed=1)
    e1.set_epoch(3); e2.set_epoch(3)
    assert list(e1) == list(e2)
    e2.set_epoch(4)
    assert list(e1) != list(e2)


def test_single_process_is_a_no_op():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    red = GradAllReducer([p])
    red.reduce()
    assert red.world == 1 and torch.equal(p.grad, torch.full((3,), 2.0))
    assert all_reduce_counts(1.0, 2.0) == [1.0, 2.0]


@pytest.mark.parametrize("n", [1, 5])
def test_arena_slot_protocol(n):
    from deepfakedetection_amd.arena import grad_dest

    params = [torch.nn.Parameter(torch.zeros(n, 3)) for _ in range(2)]
    arena = GradArena(params)
    first = grad_dest(params[0].data_ptr(), (n, 3))
    assert first is not None and first.data_ptr() == arena.slots[0].data_ptr()
    assert grad_dest(params[0].data_ptr(), (n, 3)) is None          # second write of the cycle: caller allocates
    arena.reset()
    assert grad_dest(params[0].data_ptr(), (n * 3,)) is not None
    assert grad_dest(12345, (1,)) is None
    arena.release()
    assert grad_dest(params[1].data_ptr(), (n, 3)) is None
