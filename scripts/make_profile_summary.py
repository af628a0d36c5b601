"""Turn the rocprofv3 outputs of one round into the files under profiles/.

    python scripts/make_profile_summary.py <tag> <gpurun_out/prof_TAG dir> <gpurun_out/pmc_TAG dir> <bench default json>

Inputs: `rocprofv3 --kernel-trace --stats` of `bench.py --steps 10 --warmup 2 ...` (kernel stats csv,
scripts/profile_round.sh), five `rocprofv3 --pmc` passes of `bench.py --steps 2 --warmup 1 --no-graph ...`
(FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_* | TCC_EA0_WRREQ_* | SQ/GRBM group; scripts/profile_pmc.sh), and the JSON
line of a plain `python bench.py`.
Outputs: profiles/<tag>_bench_kernel_stats.csv, _pmc_traffic.json, _pmc_sq.json, _bench_default_line.json, _summary.md."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

tag, prof_dir, pmc_dir, bench_json = sys.argv[1], Path(sys.argv[2]), Path(sys.argv[3]), Path(sys.argv[4])
# optional 5th argument: "ef:efficientformerv2_s1" / "fv:faster_vit_0_224" — counter passes of ANOTHER model of the bench line
# (scripts/profile_pmc.sh <tag> --model <name>): only profiles/<tag>_pmc_traffic_<name>.json is written (bench.py quotes it in
# models.<name>.roofline.traffic)
OTHER = sys.argv[5].split(":") if len(sys.argv) > 5 else None
ROOT = Path(__file__).resolve().parents[1]
out = ROOT / "profiles"
sys.path.insert(0, str(ROOT))
from deepfakedetection_amd.build import source_digest  # noqa: E402

SOURCE_DIGEST = source_digest()
# the digest the PMC passes themselves recorded on the GPU box (profile_pmc.sh) wins over the local one
if (pmc_dir / "csrc_sha256.txt").exists():
    SOURCE_DIGEST = (pmc_dir / "csrc_sha256.txt").read_text().strip()
try:
    import subprocess

    GIT_COMMIT = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip() or None
except Exception:  # noqa: BLE001
    GIT_COMMIT = None
FAM = collections.OrderedDict([
    # every kernel that serves a family's API calls belongs to it (the LDS-DMA product serves dfd_pwconv_fwd calls, the matrix-core
    # depthwise kernel dfd_dwconv_fwd calls): bench.py's algorithmic bytes are per API call, so bytes and time must cover one population
    ("pwconv", ("k_pw_ntw", "k_pw_ntd", "k_pw_nt<", "k_gemm_nt_dma")), ("pwconv_wgrad", ("k_pw_tnw", "k_pw_tn<")),
    ("dwconv_bwd_data", ("k_dw_bwd_data_q",)), ("dwconv_bwd_weight", ("k_dw_bwd_weight_q",)), ("dwconv_fwd", ("k_dw_fwd_q", "k_dw_fwd_mp")),
    ("act_bn_bwd", ("k_act_bn_bwd",)), ("pool", ("k_pool",)), ("bn_finalize", ("k_bn_finalize", "k_bn_bwd_finalize")),
    ("sum_partials", ("k_sum_partials", "k_sum_multi")), ("bn_bwd_reduce", ("k_bn_bwd_reduce",)), ("bn_act_apply", ("k_bn_act_apply",)),
    ("stem", ("k_stem",)), ("se_mlp", ("k_se_", "k_transpose")), ("prep_weights", ("k_prep_weights",)), ("adamw", ("k_adamw",)),
    ("pwconv_bwd_fused", ("k_pw_tnw<32, 2, 6, 0, 3, 0, 1>", "k_pw_tnw<32, 2, 6, 0, 3, 0, 2>", "k_pw_tnw<16, 2, 9, 0, 3, 0, 1>", "k_pw_tnw<16, 2, 9, 0, 3, 0, 2>")),
    # the token-mixer families of EfficientFormerV2 / FasterViT (bench.py's names)
    ("layernorm_fwd", ("k_layernorm_fwd",)), ("layernorm_bwd", ("k_layernorm_bwd",)), ("wattn_fwd", ("k_wattn_fwd",)), ("wattn_bwd", ("k_wattn_bwd",)),
    ("bgemm", ("k_bgemm",)), ("attn_gemm", ("k_attn_scores", "k_attn_apply")), ("attn_softmax_fwd", ("k_attn_softmax_fwd",)), ("attn_softmax_bwd", ("k_attn_softmax_bwd",)),
    ("bn_add_act", ("k_bn_add_act",)), ("conv3", ("k_conv3_",))])
# (the fused expand backward is an instance of k_pw_tnw: it must be matched before the plain weight-gradient family)
FAM.move_to_end("pwconv_wgrad")


def family(name):
    for f, keys in FAM.items():
        if any(k in name for k in keys):
            return f
    return "other"


stats_path = next(iter(sorted(prof_dir.glob(f"{OTHER[0] if OTHER else 'b0'}_kernel_stats.csv")) or sorted(prof_dir.glob("*kernel_stats.csv"))))
stats = list(csv.DictReader(open(stats_path)))
steps = int([r for r in stats if "k_stem_fwd" in r["Name"]][0]["Calls"])
agg = collections.OrderedDict((f, [0, 0.0]) for f in list(FAM) + ["other"])
for r in stats:
    a = agg[family(r["Name"])]
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])


def pmc(name, cols):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(pmc_dir / f"{name}_counter_collection.csv")):
        f = family(r["Kernel_Name"])
        tot[f][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[f].add(r["Dispatch_Id"])
    return tot, {f: len(v) for f, v in disp.items()}


def pmc_steps():
    seen = set()
    for r in csv.DictReader(open(pmc_dir / "FETCH_SIZE_counter_collection.csv")):
        if "k_adamw" in r["Kernel_Name"]:
            seen.add(r["Dispatch_Id"])
    return len(seen)


PMC_STEPS = pmc_steps()          # eager sizing + warm-up + timed steps of the counter runs, all eager launches
fs, nd = pmc("FETCH_SIZE", None)
ws, _ = pmc("WRITE_SIZE", None)
rd, _ = pmc("TCC_EA0_RDREQ_sum", None)
wr, _ = pmc("TCC_EA0_WRREQ_sum", None)
traffic = {}
for f in agg:
    r, w = rd[f], wr[f]
    other = max(0.0, r["TCC_EA0_RDREQ_sum"] - r["TCC_EA0_RDREQ_128B_sum"] - r["TCC_EA0_RDREQ_64B_sum"] - r["TCC_EA0_RDREQ_32B_sum"])
    ea_rd = (r["TCC_EA0_RDREQ_128B_sum"] * 128 + r["TCC_EA0_RDREQ_64B_sum"] * 64 + r["TCC_EA0_RDREQ_32B_sum"] * 32 + other * 64) / PMC_STEPS / 1e9
    ea_wr = (w["TCC_EA0_WRREQ_64B_sum"] * 64 + (w["TCC_EA0_WRREQ_sum"] - w["TCC_EA0_WRREQ_64B_sum"]) * 32) / PMC_STEPS / 1e9
    traffic[f] = {"dispatches_per_step": nd.get(f, 0) / PMC_STEPS,
                  "fetch_size_gb_per_step_raw": round(fs[f]["FETCH_SIZE"] * 1024 / PMC_STEPS / 1e9, 4),
                  "ea_read_gb_per_step": round(ea_rd, 4),
                  "write_size_gb_per_step": round(ws[f]["WRITE_SIZE"] * 1024 / PMC_STEPS / 1e9, 4),
                  "ea_write_gb_per_step": round(ea_wr, 4)}
doc = {"_about": f"HBM traffic per kernel family and training step ({OTHER[1] if OTHER else 'EfficientNet-B0'}, batch 256, 224 px, bf16). rocprofv3 --pmc in "
                 "SEPARATE passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_* | TCC_EA0_WRREQ_*) over `bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-graph --profile-steps 0` (6 eager steps; sums divided by 6). FETCH_SIZE is the raw counter "
                 "(KB -> bytes): on gfx950 it under-reports reads by 2x for every family here — TCC_EA0_RDREQ (128-byte requests x 128 B) "
                 "gives twice its bytes — so read traffic = ea_read. WRITE_SIZE agrees with TCC_EA0_WRREQ.",
       "steps": PMC_STEPS, "families": traffic,
       # provenance: bench.py refuses to quote these counters once the kernel sources differ from the ones measured
       "csrc_sha256": SOURCE_DIGEST, "git_commit": GIT_COMMIT}
if OTHER:
    doc["model"] = OTHER[1]
    (out / f"{tag}_pmc_traffic_{OTHER[1]}.json").write_text(json.dumps(doc, indent=1))
    print(f"wrote profiles/{tag}_pmc_traffic_{OTHER[1]}.json: " + ", ".join(f"{k} {v['ea_read_gb_per_step'] + v['ea_write_gb_per_step']:.2f} GB" for k, v in traffic.items() if v["dispatches_per_step"]))
    sys.exit(0)
(out / f"{tag}_pmc_traffic.json").write_text(json.dumps(doc, indent=1))
# ---- SQ / GRBM group: MFMA busy cycles, effective clock, how waves spend their life
N_SIMD = 256 * 4
sq = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(dict)
sq_file = pmc_dir / "SQ_counter_collection.csv"
if sq_file.exists():
    for r in csv.DictReader(open(sq_file)):
        f = family(r["Kernel_Name"])
        sq[f][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[f][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
sq_doc = {}
for f, c in sq.items():
    ns = sum(dur[f].values())
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                       # rocprofv3 sums the 8 XCDs
    if ns <= 0 or cyc <= 0:
        continue
    sq_doc[f] = {"ms_per_step_eager": round(ns / PMC_STEPS / 1e6, 4),
                 "effective_clock_ghz": round(cyc / ns, 3),
                 "mfma_busy_cycles_per_step": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / PMC_STEPS),
                 "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD), 4),
                 "mfma_mops_bf16_per_step": round(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] / PMC_STEPS),
                 "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD), 4),
                 "waves_per_simd": round(c["SQ_WAVE_CYCLES"] * 4.0 / (cyc * N_SIMD), 2),
                 "wave_parked_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c["SQ_WAVE_CYCLES"] else None}
(out / f"{tag}_pmc_sq.json").write_text(json.dumps(
    {"_about": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES "
               "SQ_WAIT_ANY GRBM_GUI_ACTIVE over the eager bench (same command as the traffic passes), summed per kernel family. "
               "effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (reads high for dispatches under ~0.3 ms, per the "
               "microarchitecture guide); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); valu_busy_frac = "
               "SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024) (quad-cycle counter); wave_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES "
               "(share of wave lifetime spent in s_waitcnt / s_barrier).",
     "steps": PMC_STEPS, "families": sq_doc}, indent=1))
shutil.copy(stats_path, out / f"{tag}_bench_kernel_stats.csv")
line = json.loads(bench_json.read_text().strip().splitlines()[-1])
(out / f"{tag}_bench_default_line.json").write_text(json.dumps(line) + "\n")
live = {k["kernel"]: k for k in line["kernels"]}
L = [f"# Build {tag} — rocprofv3 kernel stats next to bench.py's live numbers\n",
     "Commands (MI355X, one GPU):\n",
     f"* `python bench.py` → `profiles/{tag}_bench_default_line.json` ({line['value']} images/sec, {line['ms_per_step']} ms/step, "
     f"launch: {line['config']['launch']}; f32 eval forward {line['config'].get('eval_f32_images_per_sec_per_gpu')} images/sec).",
     f"* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 "
     f"--eval-steps 0` → `profiles/{tag}_bench_kernel_stats.csv` ({steps} steps executed in that process; the table divides by {steps}).",
     f"* five `rocprofv3 --pmc` passes (one counter group each, `scripts/profile_pmc.sh`) → `profiles/{tag}_pmc_traffic.json`, "
     f"`profiles/{tag}_pmc_sq.json`; `scripts/make_profile_summary.py` made this file.\n",
     "Algorithmic bytes are SURVEY §8(d)'s (every tensor of the layer once; the engine's extra fusion operands — the second operand of the "
     "BN-backward map, residuals — are NOT counted); `traffic ÷ algorithmic` therefore includes them.\n",
     "| family | launches/step | avg launch µs (rocprofv3) | ms/step (rocprofv3) | ms/step (bench.py live) | §8(d) GB/s (rocprofv3 time) | frac of 8 TB/s | HBM read GB/step | HBM write GB/step | traffic ÷ §8(d) bytes | MFMA busy | VALU busy | waves/SIMD | wave parked | eff. clock GHz |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
total = 0.0
for f, (calls, ns) in agg.items():
    if not calls:
        continue
    lv = live.get(f)
    t = traffic[f]
    ratio = gbps = frac = ""
    if lv and lv["GBps"] > 100 and lv["ms_per_step"]:
        algo = lv["GBps"] * lv["ms_per_step"] / 1e3                  # GB per step, SURVEY 8(d) accounting
        ratio = f"{(t['ea_read_gb_per_step'] + t['ea_write_gb_per_step']) / algo:.2f}"
        g = algo / (ns / steps / 1e9) / 1e0
        gbps, frac = f"{g:.0f}", f"{g / 8000.0:.3f}"
    q = sq_doc.get(f, {})
    total += ns / steps / 1e6
    L.append(f"| {f} | {calls / steps:.1f} | {ns / calls / 1e3:.1f} | {ns / steps / 1e6:.3f} | {lv['ms_per_step'] if lv else ''} | "
             f"{gbps} | {frac} | {t['ea_read_gb_per_step']:.2f} | {t['ea_write_gb_per_step']:.2f} | {ratio} | "
             f"{q.get('mfma_busy_frac', '')} | {q.get('valu_busy_frac', '')} | {q.get('waves_per_simd', '')} | {q.get('wave_parked_frac', '')} | "
             f"{q.get('effective_clock_ghz', '')} |")
r = line["roofline"]
top = agg[r["kernel"]]
tot_rd = sum(t["ea_read_gb_per_step"] for t in traffic.values())
tot_wr = sum(t["ea_write_gb_per_step"] for t in traffic.values())
L.append(f"\nSum of kernel durations: {total:.2f} ms per step, {sum(c for c, _ in agg.values()) / steps:.0f} launches per step "
         f"(hipGraph step: {line['ms_per_step']} ms).  HBM traffic of the whole step: {tot_rd:.1f} GB read + {tot_wr:.1f} GB written "
         f"= {tot_rd + tot_wr:.1f} GB ({(tot_rd + tot_wr) / line['config']['per_gpu_batch'] * 1e3:.0f} MB per image).")
L.append(f"\nDominant family (`roofline` of the bench line): **{r['kernel']}**, bound hbm, {r['achieved']} GB/s of {r['peak']} "
         f"(frac {r['frac']}); live average launch {r['avg_launch_us']} µs vs rocprofv3 {top[1] / top[0] / 1e3:.1f} µs; "
         f"traffic {r['traffic'] / 1e6 if r['traffic'] else float('nan'):.1f} MB per launch against {r['avg_launch_bytes'] / 1e6:.1f} MB algorithmic.")
cb = line.get("cpu_baseline") or {}
L.append(f"\nCPU baseline of the same line: {cb.get('value')} images/sec on {cb.get('cores')} cores ({cb.get('kind')}; {cb.get('sample')}).")
(out / f"{tag}_summary.md").write_text("\n".join(L) + "\n")
print("\n".join(L))
