"""Headline benchmark: EfficientNet-B0 train images/sec @224x224 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = the reference's hot loop body (trainers/efficientnet.py:290-309) on one
synthetic batch already resident in HBM: bf16-autocast forward, label-smoothed CE,
backward, [gradient all-reduce,] AdamW — every kernel from libdfd_hip.so.  The step is
captured into one hipGraph when capture succeeds (launch: "hipgraph"), otherwise it runs
eagerly (launch: "eager").  Rank 0 prints ONE JSON line (see README/DESIGN for fields):
  roofline      — dominant kernel family of the step, measured live with HIP events on the
                  stream the kernels run on, against the HBM (or MFMA) peak of the guide;
  cpu_baseline  — the CPU oracle (same op sequence through ATen/oneDNN CPU kernels, f32,
                  channels_last) timed on this host's cores on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config: 256)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--model", default="efficientnet",
                    help="'efficientnet' (BASELINE config 2; see --variant / --flavour), an EfficientFormerV2 name such as "
                         "efficientformerv2_s1 (BASELINE config 3) or a FasterViT name such as faster_vit_0_224 (config 5)")
    ap.add_argument("--variant", default="b0")
    ap.add_argument("--flavour", default="timm")
    ap.add_argument("--fp8-weights", action="store_true",
                    help="FasterViT: qkv / proj / fc1 / fc2 weights as OCP MX fp8 on the block-scaled fp8 MFMA (BASELINE config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dp-mode", choices=("split", "overlap"), default="split",
                    help="N>1: 'split' = hipGraph(fwd+bwd) | RCCL all-reduce | hipGraph(AdamW); "
                         "'overlap' = eager step, buckets all-reduced from backward hooks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32, help="SURVEY 8(d): N = 32 with the ratio to the GPU batch stated")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU steps (median)")
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-timeout", type=int, default=240, help="seconds granted to the CPU baseline child process")
    ap.add_argument("--extra-models", default="efficientformerv2_s1,faster_vit_0_224",
                    help="N = 1, default workload only: after the timed region, also measure these models (batch 256, bf16) in "
                         "child processes and report them under `models` ('' to skip)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) measure the CPU oracle and print its JSON")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--eval-steps", type=int, default=10, help="f32 eval-mode forward passes timed after the train steps (0: skip)")
    return ap.parse_args()


def usable_cores() -> int:
    """CPU cores this process may actually use: affinity mask and cgroup quota, not the host's count
    (a GPU box exposes hundreds of host CPUs to os.cpu_count() while the container owns a few)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


def model_label(args) -> str:
    if args.model.startswith("efficientformer"):
        return "EfficientFormerV2-" + args.model.rsplit("_", 1)[-1].upper()
    if args.model.startswith("faster_vit"):
        return "FasterViT-" + args.model.split("_")[2]
    return f"EfficientNet-{args.variant.upper()}"


def cpu_baseline_measure(args) -> dict:
    """Oracle train step on the host cores; bounded sample, ~10-30 s.  Runs in a CHILD process
    (see cpu_baseline) so that it can be time-boxed and never shares threads with the GPU run."""
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    if args.model.startswith("efficientformer"):
        from oracle.efformer_ref import EfficientFormerV2Ref, train_step_ref, variant_of

        model = EfficientFormerV2Ref(variant_of(args.model), args.classes, args.size).to(memory_format=torch.channels_last)
    elif args.model.startswith("faster_vit"):
        from oracle.fastervit_ref import FasterViTRef, train_step_ref, variant_of

        model = FasterViTRef(variant_of(args.model), args.classes, args.size).to(memory_format=torch.channels_last)
    else:
        from oracle.effnet_ref import EfficientNetRef, train_step_ref

        model = EfficientNetRef(args.variant, args.flavour, args.classes).to(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.cpu_batch, 3, args.size, args.size, generator=g).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, args.classes, (args.cpu_batch,), generator=g)
    for _ in range(max(1, args.cpu_warmup)):
        train_step_ref(model, opt, x, y)
    times = []
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        train_step_ref(model, opt, x, y)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(args.cpu_batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "batch": args.cpu_batch, "gpu_batch_over_cpu_batch": round(args.batch / args.cpu_batch, 3),
            "sample": f"oracle {model_label(args)} f32 channels_last train step (fwd+CE+bwd+AdamW), "
                      f"batch {args.cpu_batch} @{args.size}px (the GPU step's batch is {args.batch}: ratio {args.batch / args.cpu_batch:g}), "
                      f"median of {args.cpu_steps} timed steps after {max(1, args.cpu_warmup)} warm-up steps"}


def cpu_baseline(args) -> dict:
    """Time-boxed: the measurement runs as `bench.py --cpu-baseline-only` in a child process."""
    import subprocess

    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--model", args.model, "--variant", args.variant,
           "--flavour", args.flavour,
           "--classes", str(args.classes), "--size", str(args.size), "--batch", str(args.batch), "--cpu-batch", str(args.cpu_batch),
           "--cpu-steps", str(args.cpu_steps), "--cpu-warmup", str(args.cpu_warmup)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""          # the child is a pure CPU program
    env["OMP_NUM_THREADS"] = str(usable_cores())
    try:
        proc = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        for line in reversed(proc.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        note = f"child exited {proc.returncode}: {proc.stderr.strip()[-200:]}"
    except subprocess.TimeoutExpired:
        note = f"did not finish within {args.cpu_timeout} s"
    return {"value": None, "unit": "images/sec", "cores": usable_cores(), "kind": "port", "sample": f"not measured ({note})"}


def pmc_traffic(family: str, args) -> tuple[int | None, dict]:
    """HBM bytes PER STEP of a kernel family from the committed rocprofv3 --pmc passes (counters cannot be read from
    inside this process): newest profiles/*pmc_traffic.json, which records GB per training step of exactly this workload.
    The caller divides by ITS launch count (API calls per step), the same population its algorithmic bytes are divided by —
    rocprofv3 counts kernel dispatches, which differ where one call is several kernels or none (VERDICT r3 "what's weak" 7).
    Returns (bytes or None, provenance).  None when the file is absent, the workload differs, or the file was collected
    from OTHER kernel sources than the ones this run was built from (the file records a digest of csrc/ + include/;
    build.source_digest() recomputes it) — a stale counter file must not dress up new kernels."""
    if (args.batch, args.size) != (256, 224) or args.fp8_weights:
        return None, {"file": None, "reason": "counter passes exist for batch 256 at 224 px (bf16) only"}
    if (args.model, args.variant, args.flavour) == ("efficientnet", "b0", "timm"):
        pattern = "*pmc_traffic.json"
    elif args.model != "efficientnet":
        pattern = f"*pmc_traffic_{args.model}.json"                # the other models of the bench line (scripts/profile_pmc.sh --model ...)
    else:
        return None, {"file": None, "reason": "counter passes exist for the bench line's three models only"}
    files = sorted((ROOT / "profiles").glob(pattern))
    if not files:
        return None, {"file": None, "reason": f"no profiles/{pattern}"}
    from deepfakedetection_amd.build import source_digest

    src = {"file": f"profiles/{files[-1].name}"}
    try:
        doc = json.loads(files[-1].read_text())
        src["git_commit"] = doc.get("git_commit")
        src["csrc_sha256"] = doc.get("csrc_sha256")
        if doc.get("csrc_sha256") != source_digest():
            src["reason"] = "stale: kernel sources changed since these counters were collected"
            return None, src
        fam = doc["families"][family]
        return int((fam["ea_read_gb_per_step"] + fam["ea_write_gb_per_step"]) * 1e9), src
    except (KeyError, ValueError, ZeroDivisionError) as exc:
        src["reason"] = f"unreadable ({type(exc).__name__})"
        return None, src


def extra_model_line(name: str, args) -> dict:
    """`bench.py --model name` (batch 256, bf16, hipGraph) in a child process, AFTER this process's timed region:
    BASELINE.json's metric names EfficientFormerV2-S1 next to EfficientNet-B0 (and config 5 names FasterViT-0)."""
    import subprocess

    cmd = [sys.executable, str(ROOT / "bench.py"), "--model", name, "--batch", str(args.batch), "--size", str(args.size),
           "--classes", str(args.classes), "--steps", str(min(args.steps, 20)), "--warmup", str(min(args.warmup, 5)),
           "--no-cpu-baseline", "--eval-steps", "0", "--extra-models", ""]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        proc = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        for line in reversed(proc.stdout.strip().splitlines()):
            if line.startswith("{"):
                d = json.loads(line)
                return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                        "dtype": d["dtype"], "per_gpu_batch": d["config"]["per_gpu_batch"], "launch": d["config"]["launch"],
                        "workload": d["config"]["workload"], "roofline": d["roofline"],
                        "top_kernels": d["kernels"][:6]}
        return {"value": None, "note": f"child exited {proc.returncode}: {proc.stderr.strip()[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "note": "did not finish within 300 s"}


def progress(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_measure(args)), flush=True)
        return
    from deepfakedetection_amd import kernels as K
    from deepfakedetection_amd.dp import GradAllReducer, broadcast_module_state, init_distributed, rccl_log_request, rccl_log_summary
    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    rccl_log = rccl_log_request("bench") if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0 else None
    rank, local_rank, world = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # one rank per GPU; the modulo only matters when a rehearsal runs several ranks on a one-GPU box
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)

    torch.manual_seed(1)
    if args.model.startswith("efficientformer"):
        from deepfakedetection_amd.efficientformer_v2 import build_efficientformer_v2

        model = build_efficientformer_v2(args.model, args.classes, args.size).to(device).train()
        workload = f"{model_label(args)} (timm 1.0.20 architecture)"
    elif args.model.startswith("faster_vit"):
        from deepfakedetection_amd.fastervit import build_fastervit

        model = build_fastervit(args.model, args.classes, fp8_weights=args.fp8_weights).to(device).train()
        workload = f"{model_label(args)} (fastervit 1.0.0 architecture, DropPath 0.2{', MX fp8 Linear weights' if args.fp8_weights else ''})"
    else:
        model = HipEfficientNet(args.variant, args.flavour, args.classes).to(device).train()
        workload = f"EfficientNet-{args.variant} ({args.flavour} flavour)"
    broadcast_module_state(model)
    opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2, grad_scale=1.0 / world)
    reducer = GradAllReducer(model.parameters(), arena=opt.arena) if world > 1 else None
    crit = HipCrossEntropyLoss(label_smoothing=0.1)
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(device).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, args.classes, (args.batch,), generator=g).to(device)
    loss_box: list[torch.Tensor] = [torch.zeros((), device=device)]

    overlap = reducer is not None and args.dp_mode == "overlap"
    if overlap:
        reducer.attach()

    def fwd_bwd() -> None:
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(x), y)
        if overlap:
            reducer.arm()
        loss.backward()
        loss_box[0] = loss.detach()

    def fwd_bwd_plain() -> None:           # the same without arming the hooks (profiling on one rank)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(x), y)
        loss.backward()

    gloo_on_gpu = world > 1 and dist.get_backend() == "gloo"     # one-GPU rehearsal only

    def exchange() -> None:
        if gloo_on_gpu and not overlap:
            torch.cuda.synchronize()        # gloo stages CUDA tensors through the host: hand it finished data
        if reducer is not None:
            if overlap:
                reducer.finish()
            else:
                reducer.reduce()

    def step_body() -> None:
        fwd_bwd()
        exchange()
        opt.step()

    # eager warm-up: sizes the scratch buffers, builds the optimizer state / tables
    for _ in range(3):
        step_body()
    torch.cuda.synchronize()

    # Launch modes.  N = 1: the whole step is ONE hipGraph.  N > 1 ("split"): graph_step.GraphedTrainStep — the very
    # object the trainers drive (trainers/efficientnet.make_stepper): graph(zero_grad + forward + loss + backward), then
    # the RCCL all-reduce of the flat gradient arena in bucket-sized chunks OUTSIDE of graph capture, then graph(AdamW).
    # ("overlap" runs eagerly so that the backward hooks can fire.)
    launch = "eager"
    graph_a = None
    stepper = None
    if not args.no_graph and not overlap:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if reducer is None:
                opt.prepare_step()
                graph_a = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_a, capture_error_mode="thread_local"):
                    fwd_bwd()
                    opt.step()
                launch = "hipgraph"
            else:
                from deepfakedetection_amd.graph_step import GraphedTrainStep

                stepper = GraphedTrainStep(model, crit, opt, accum_steps=1, use_amp=True, eager_cycles=1, reducer=reducer)
                for _ in range(3):          # eager cycle, first sight of the shape, capture + first replay
                    loss_box[0] = stepper.micro_batch(x, y, first=True, last=True)
                    stepper.optimizer_step()
                torch.cuda.synchronize()
                if stepper.failed or stepper.step_graph is None:
                    raise RuntimeError("GraphedTrainStep fell back to eager")
                coll = "rccl" if dist.get_backend() == "nccl" else dist.get_backend()
                if stepper.cut_modules and stepper.segmented_replays:
                    launch = (f"hipgraph(fwd) | {len(stepper.cut_modules) + 1} x [hipgraph(bwd segment) -> {coll} all-reduce of the buckets it "
                              f"completed, overlapped with the next segment] | hipgraph(adamw)")
                else:
                    launch = f"hipgraph(fwd+bwd) | {coll} all-reduce | hipgraph(adamw)"
        except Exception as exc:  # noqa: BLE001 - any capture failure means: measure eagerly
            if rank == 0:
                import traceback

                traceback.print_exc()
                print(f"[bench] hipGraph capture failed ({type(exc).__name__}); running eagerly", file=sys.stderr)
            graph_a = stepper = None
            torch.cuda.synchronize()
    elif overlap:
        launch = "eager, all-reduce overlapped with backward"

    def run_step() -> None:
        if graph_a is not None:
            opt.prepare_step()
            graph_a.replay()
        elif stepper is not None:
            loss_box[0] = stepper.micro_batch(x, y, first=True, last=True)
            stepper.optimizer_step()
        else:
            step_body()

    if rank == 0:
        progress(f"launch mode: {launch}; warm-up {args.warmup} steps, timing {args.steps} steps")
    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss_box[0])

    # evaluate()'s forward (reference trainers/efficientnet.py:237-262: eval mode, f32, no autocast), same batch
    eval_ips = None
    if args.eval_steps > 0:
        model.eval()
        with torch.inference_mode():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            for _ in range(args.eval_steps):
                model(x)
            torch.cuda.synchronize()
            eval_ips = args.batch * args.eval_steps / (time.perf_counter() - t_e)
        model.train()
    in_sync = None
    if world > 1:
        # replicas started from rank 0's weights and applied the same averaged gradients
        chk = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        parts = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(parts, chk)
        in_sync = all(bool(torch.equal(parts[0], q)) for q in parts)

    # ---- live per-kernel measurement (eager, HIP events on the compute stream)
    roofline, breakdown = None, []
    if rank == 0 and args.profile_steps > 0:
        sink: list = []

        def local_step() -> None:          # rank 0 alone: no collective in here
            fwd_bwd_plain()
            opt.step()

        local_step()
        torch.cuda.synchronize()
        K.set_profile_sink(sink)
        for _ in range(args.profile_steps):
            local_step()
        torch.cuda.synchronize()
        K.set_profile_sink(None)
        agg: dict[str, list[float]] = {}
        for name, nbytes, flops, e0, e1, b8d in sink:
            a = agg.setdefault(name, [0.0, 0.0, 0.0, 0, 0.0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += nbytes
            a[2] += flops
            a[3] += 1
            a[4] += b8d
        total_t = sum(a[0] for a in agg.values())
        # GBps: SURVEY section 8(d) algorithmic bytes (kernels._bytes_8d); GBps_incl_fusion_operands: every tensor the
        # call touches, i.e. including what the engine's own fusions add (second affine2 operand, residual)
        for name, (t, b, f, n, b8) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            breakdown.append({"kernel": name, "launches_per_step": n // args.profile_steps,
                              "ms_per_step": round(t / args.profile_steps * 1e3, 4),
                              "share": round(t / total_t, 4), "GBps": round(b8 / t / 1e9, 1),
                              "frac_of_hbm_peak": round(b8 / t / 1e9 / HBM_PEAK_GBS, 4),
                              "GBps_incl_fusion_operands": round(b / t / 1e9, 1), "TFLOPs": round(f / t / 1e12, 2)})
        # dominant family by time; every family in this step is HBM-bound by arithmetic
        # intensity (all 1x1 layers < 312 flop/B), so bound = "hbm"
        top = breakdown[0]
        t, b, f, n, b8 = agg[top["kernel"]]
        traffic_step, traffic_src = pmc_traffic(top["kernel"], args)
        calls_per_step = n / args.profile_steps
        traffic = int(traffic_step / calls_per_step) if traffic_step is not None else None
        roofline = {"kernel": top["kernel"], "bound": "hbm", "achieved": round(b8 / t / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(b8 / t / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    # both per step and over the same launches: traffic / algorithmic is the re-read factor
                    "traffic_per_step": traffic_step, "algorithmic_bytes_per_step": int(b8 / args.profile_steps),
                    "calls_per_step": round(calls_per_step, 2),
                    "traffic_over_algorithmic": round(traffic_step / (b8 / args.profile_steps), 3) if traffic_step else None,
                    "bytes": "SURVEY 8(d): M*(K+Nout)*2 + K*Nout*2 per 1x1 pass; N*C*(Hin*Win+Hout*Wout)*2 + k*k*C*4 per depthwise forward",
                    "achieved_incl_fusion_operands": round(b / t / 1e9, 1),
                    "avg_launch_us": round(t / n * 1e6, 2), "avg_launch_bytes": int(b8 / n),
                    "share_of_step": top["share"], "mfma_tflops": round(f / t / 1e12, 2),
                    "mfma_frac": round(f / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        pw = [agg[k] for k in ("pwconv", "pwconv_wgrad", "pwconv_bwd_fused") if k in agg]
        if pw:
            tpw, bpw, fpw = sum(a[0] for a in pw), sum(a[4] for a in pw), sum(a[2] for a in pw)
            # every 1x1 launch of the step (forward, data gradient, weight gradient, the fused expand backward) as one family
            roofline["pointwise_family"] = {"achieved": round(bpw / tpw / 1e9, 1), "frac": round(bpw / tpw / 1e9 / HBM_PEAK_GBS, 4),
                                            "ms_per_step": round(tpw / args.profile_steps * 1e3, 3),
                                            "mfma_frac": round(fpw / tpw / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        dw = [agg[k] for k in ("dwconv_fwd", "dwconv_bwd_data", "dwconv_bwd_weight") if k in agg]
        if dw:
            tdw, bdw = sum(a[0] for a in dw), sum(a[4] for a in dw)
            roofline["depthwise_family"] = {"achieved": round(bdw / tdw / 1e9, 1), "frac": round(bdw / tdw / 1e9 / HBM_PEAK_GBS, 4),
                                            "ms_per_step": round(tdw / args.profile_steps * 1e3, 3)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        line = {
            "metric": f"train images/sec @{args.size}^2 ({model_label(args)})", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 (MX fp8 e4m3 Linear forward)" if args.fp8_weights else "bf16", "data": "synthetic",
            "config": {"workload": f"{workload} {args.size}x{args.size} train step: "
                                   f"bf16 fwd + label-smoothed CE + bwd + AdamW, random-init weights, {args.classes} classes",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "launch": launch, "final_loss": round(final_loss, 4), "replicas_in_sync": in_sync,
                       "rccl": rccl_log_summary(rccl_log) if world > 1 else None,
                       "eval_f32_images_per_sec_per_gpu": round(eval_ips, 1) if eval_ips else None},
            "roofline": roofline,
            "kernels": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            progress(f"{value:.0f} images/sec measured; timing the CPU oracle on {usable_cores()} host cores (bounded)")
            line["cpu_baseline"] = cpu_baseline(args)
        if world == 1 and args.extra_models not in ("", "none") and (args.model, args.variant, args.flavour) == ("efficientnet", "b0", "timm"):
            line["models"] = {}
            for name in [m for m in args.extra_models.split(",") if m and m != "none"]:
                progress(f"extra model {name} (child process, outside the timed region)")
                line["models"][name] = extra_model_line(name, args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
