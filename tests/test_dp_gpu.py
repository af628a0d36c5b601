"""N > 1 path on the GPU: two ranks share cuda:0 and exchange over gloo (RCCL refuses two ranks
on one device; on a real node the same code runs one rank per GPU over RCCL).  Checks that
  * the all-reduced gradient arena equals the sum of the two ranks' local arenas (one-shot path and
    the hook-driven overlapped path),
  * after AdamW steps with grad_scale = 1/world the replicas hold identical parameters."""

from __future__ import annotations

import datetime
import os
import socket
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, mode: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from deepfakedetection_amd.dp import GradAllReducer, broadcast_module_state
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    # a failing rank must never leave its peer blocked in a collective: short timeout, hard exit
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        torch.manual_seed(10 + rank)                                 # different init: broadcast must fix it
        model = HipEfficientNet("b0", "timm", 2).to(dev).train()
        broadcast_module_state(model)
        opt = HipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-2, grad_scale=1.0 / world)
        red = GradAllReducer(model.parameters(), bucket_bytes=2 << 20, arena=opt.arena)
        if mode == "hooks":
            red.attach()
        crit = HipCrossEntropyLoss(label_smoothing=0.1)
        g = torch.Generator().manual_seed(50 + rank)                 # each rank its own shard
        x = torch.randn(8, 3, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 2, (8,), generator=g).to(dev)
        for step in range(2):
            opt.zero_grad(set_to_none=True)
            # the engine draws its dropout uniforms from a device-resident Philox state that advances every
            # forward pass; the two forwards compared below take the same explicit uniforms instead
            u = torch.rand(8, 1280, generator=torch.Generator().manual_seed(1000 + step)).to(dev)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = crit(model(x, dropout_u=u), y)
            if mode == "hooks":
                # reference: a hook-free backward of the same step (same dropout mask)
                loss.backward()
                torch.cuda.synchronize()
                local = opt.arena.flat.clone()
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = crit(model(x, dropout_u=u), y)
                red.arm()
                loss.backward()
                red.finish()
            else:
                loss.backward()
                torch.cuda.synchronize()
                local = opt.arena.flat.clone()
                red.reduce()
            torch.cuda.synchronize()
            parts = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(parts, local)
            want = parts[0] + parts[1]
            got = opt.arena.flat
            assert opt.arena.holds_all_grads()
            err = (got - want).abs().max().item()
            assert err <= 1e-5 * max(1.0, want.abs().max().item()), f"{mode}: reduced arena differs from the sum ({err})"
            opt.step()
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        parts = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        assert torch.equal(parts[0], parts[1]), f"{mode}: replicas diverged"
        if mode == "hooks":
            assert red.launched_early > 0, "no bucket left from inside backward"
    except BaseException:
        traceback.print_exc()
        os._exit(1)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["oneshot", "hooks"])
def test_two_ranks_one_gpu(mode):
    mp.spawn(_worker, args=(2, _free_port(), mode), nprocs=2, join=True)


def _graph_worker(rank: int, world: int, port: int, accum: int) -> None:
    """The trainers' DP path under hipGraph replay (trainers.efficientnet.make_stepper(world > 1) -> GraphedTrainStep
    with a reducer): graph(zero_grad+fwd+bwd) -> all-reduce of the arena -> graph(AdamW) must leave bit for bit the
    parameters the eager DP loop (hook-driven, overlapped exchange) leaves, over several optimizer cycles with
    gradient accumulation, and the replicas must stay identical."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GRAPH_STEP="1")
    from deepfakedetection_amd.dp import GradAllReducer, broadcast_module_state
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
    from deepfakedetection_amd.trainers.efficientnet import make_stepper

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(70 + rank)                 # each rank its own shard
        cycles = 5
        batches = [(torch.randn(8, 3, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last),
                    torch.randint(0, 2, (8,), generator=g).to(dev)) for _ in range(cycles * accum)]

        def run(graph: bool, segments: bool = True):
            os.environ["DFD_DP_SEGMENTS"] = "1" if segments else "0"
            torch.manual_seed(10 + rank)                             # different init: broadcast must fix it
            model = HipEfficientNet("b0", "timm", 2).to(dev).train()
            broadcast_module_state(model)
            opt = HipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-2, grad_scale=1.0 / world)
            red = GradAllReducer(model.parameters(), bucket_bytes=2 << 20, arena=opt.arena)
            red.attach()
            step = make_stepper(model, HipCrossEntropyLoss(0.1), opt, accum_steps=accum, use_cuda=True, world=world, reducer=red)
            assert step is not None, "make_stepper must serve world > 1"
            if not graph:
                step.failed = True                                   # the object's eager path: armed hooks, overlapped exchange
            for i, (x, y) in enumerate(batches):
                step.micro_batch(x, y, first=i % accum == 0, last=(i + 1) % accum == 0)
                if (i + 1) % accum == 0:
                    step.optimizer_step()
            torch.cuda.synchronize()
            red.detach()
            return model, step, red

        m_e, _, red_e = run(False)
        m_g, step, red_g = run(True)
        m_1, step1, _ = run(True, segments=False)
        assert red_e.launched_early > 0, "eager DP: no bucket left from inside backward"
        assert not step.failed and step.step_graph is not None and step.replays == (cycles - 1) * accum, step.replays
        # the replayed backward that completes a cycle ran as segments, and buckets left between them (overlap under replay)
        assert len(step.cut_modules) >= 2 and step.segmented_replays == cycles - 1, (len(step.cut_modules), step.segmented_replays)
        assert red_g.launched_early >= (cycles - 1) * 2, red_g.launched_early
        assert not step1.cut_modules and step1.segmented_replays == 0 and step1.replays == (cycles - 1) * accum
        for (name, a), (_, b), (_, c) in zip(m_e.state_dict().items(), m_g.state_dict().items(), m_1.state_dict().items()):
            assert torch.equal(a, b), f"graphed (segmented) DP differs from eager DP at {name}"
            assert torch.equal(a, c), f"graphed (single graph) DP differs from eager DP at {name}"
        flat = torch.cat([p.detach().flatten() for p in m_g.parameters()])
        parts = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        assert torch.equal(parts[0], parts[1]), "replicas diverged under graph replay"
    except BaseException:
        traceback.print_exc()
        os._exit(1)
    dist.destroy_process_group()


@pytest.mark.parametrize("accum", [1, 2])
def test_graphed_dp_trainer_step_is_bitwise_the_eager_dp_step(accum):
    mp.spawn(_graph_worker, args=(2, _free_port(), accum), nprocs=2, join=True)
