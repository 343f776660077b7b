#!/usr/bin/env python3
"""Headline benchmark of the MI355X depth hot path (BASELINE.json: "ref-views/sec at 1152x1536 N=5 D=192 4-stage;
achieved HBM GB/s vs peak").

    python bench.py --gpus 1 --steps 20 --warmup 3          # 20 steps x 96 reference views
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

One "step" = one batch of `--views-per-step` (default 96) reference views, each pushed through the whole 4-stage cascade with
batch 1 per forward call like the reference's test.py loop (features in HBM -> refined depth + confidence): hypothesis
scheduling, fused warp + group-wise correlation + visibility weighting for the 4 source views, the 3D-conv regulariser and the
depth head of every stage.  Consecutive reference views rotate over `--input-sets` (default 4) distinct synthetic input sets
(4 x 531 MB of fp32 features > the 256 MiB Infinity Cache), so the default 20 steps time 1920 reference views / >= 2 s and no view
finds its inputs cache-resident by construction.  `value` = reference views per second over the timed steps.
Workload = BASELINE.json configs[1]: 1152x1536, V=5, numdepth 192 (-> cascade ndepths 32/16/8/4),
fp32 features, all-"Normal" regularisers, synthetic features / cameras and seeded random weights with randomised BatchNorm
statistics (no dataset or checkpoint exists offline).

Arithmetic: warp, correlation, visibility, heads and every accumulation in fp32; the precision policy of the stages is the PRODUCT
DEFAULT "auto" (module.DEFAULT_CASCADE_POLICY, round 6): the head reads the depth range it is handed and runs "f16mix" on every stage while
depth_max / depth_min stays below half the ratio at which the inverse-depth schedule degenerates (this workload: 2.2 against 12.6) - fp16
activation tensors; fp16 hi + lo weights = two MFMA terms per product on the 8- / 16-channel layers, one fp16 term on conv4..conv7 and in the
visibility CNN; fp16 source windows and kept correlations in the gather: no narrower than the bf16 autocast the reference's own GPU path runs
these layers under (test.py:250) - and round 5's "stagemix" beyond it (coarse stages fp32-equivalent: split-bf16 regulariser and visibility CNN,
exact gather), reported here as `exact_coarse_mode`.  `--conv-precision bf16x3` selects the fp32-equivalent format on every stage.  `parity` = this run's refined depth against the fp32 CPU oracle on the
same inputs (bar 1e-3).

The reference views of a step are issued round-robin on `--streams` HIP streams (default 4 since round 5 - graph replay and the default
policy's launch-bound coarse stages moved the optimum from 3: 606 vs 595 ref-views/s, three alternations on one box): views are independent, so the small
latency-bound launches of one view's coarse stages overlap the large launches of another's fine stages; the single-stream
figure (one view at a time, the reference's loop) is reported in `latency`, the per-kernel profile behind `roofline` is
single-stream too.  Issue (round 5): one hipGraph replay per reference view (`CascadeDepthHead.capture`, one graph per (stream,
input set); `--issue eager` = the ~60 launches of a view issued from Python) - the same device work; the host needs ~1.1 ms per view to
issue it eagerly, which a device path below ~1.3 ms per view no longer hides.

N > 1: one process per GPU; reference views are independent, so every rank runs its own stream of reference views
(data parallel over reference views, no data-path collective; "scaling": "weak").  The view-sharded latency mode
(source views over ranks, partial cost volumes combined per stage over RCCL) is timed afterwards on
BASELINE configs[2]'s shape (V = 10) and reported in the extra `view_sharded` object.

Rank 0 prints ONE JSON line.  Extra objects: `training_step` (N = 1: forward + backward of each cascade stage through the native
training kernels at DTU-training-like sizes, outside the timed region), `roofline` (dominant kernel, HIP-event timing on the launch stream) and
`cpu_baseline` (the oracle - a CPU restatement of the reference path - timed on this host's cores, N=1 only), `fp32_equivalent_mode`
(N = 1: the same weights, inputs and loop with the regularisers in "bf16x3", outside the timed headline: both modes in one line),
`uniform_f16mix_mode` (N = 1, round 5: the same with rounds 3-4's uniform fp16 format - what the default policy's exact coarse stages cost),
`shipped` (N = 1, round 5: the SHIPPED regulariser mix - stage-1 transformer + PE3D, config/mvsformer++.json:86-113 - on the same inputs:
ref-views/s and its own parity against the oracle, outside the timed headline).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = "DTU 1152x1536 N=5 numdepth=192 4-stage cascade (ndepths 32/16/8/4, all-Normal CostRegNet/CostRegNet3D)"
ARGS = {"base_ch": [8, 8, 8, 8], "depth_type": ["ce"] * 4, "fusion_type": "cnn", "cost_reg_type": ["Normal"] * 4,
        "ndepths": [32, 16, 8, 4], "depth_interals_ratio": [4.0, 2.67, 1.5, 1.0], "inverse_depth": True}
TMP = [5.0, 5.0, 5.0, 1.0]
ALGO_BYTES_PER_VIEW = 5.03e9
# `dtype` of the JSON line = the arithmetic the regularisers (the only reduced-precision part) compute in; warp / correlation /
# visibility maps / heads / every accumulation are fp32 in all of them ("f32 (...)" mislabelled fp16 operands)
DTYPE_TEXT = {
    "stagemix": "fp16 operands/storage, fp32 accumulate (fine stages 3-4: fp16 U-Net tensors, source windows and kept correlations, 1-2 fp16 MFMA "
                "terms; coarse stages 1-2: split-bf16 operands x3 terms = fp32-equivalent, exact gather)",
    "f16mix": "fp16 operands/storage, fp32 accumulate (every stage; 1-2 fp16 MFMA terms)",
    "f16x2": "fp16 operands/storage, fp32 accumulate (every stage; 2 fp16 MFMA terms)",
    "f16": "fp16 operands/storage, fp32 accumulate (every stage; 1 fp16 MFMA term)",
    "bf16x3": "split-bf16 operands x3 MFMA terms (fp32-equivalent activations), fp32 accumulate",
    "fp32": "f32"}


# stage-1 regulariser of the shipped checkpoints (config/mvsformer++.json:86-113); reported beside the all-"Normal" headline
SHIPPED = {"cost_reg_type": ["PureTransformerCostReg", "Normal", "Normal", "Normal"], "use_pe3d": True,
           "transformer_config": [{"base_channel": 8, "mid_channel": 64, "num_heads": 4, "down_rate": [2, 4, 4], "mlp_ratio": 4,
                                   "layer_num": 6, "drop": 0.0, "attn_drop": 0.0, "position_encoding": True, "attention_type": "FLASH2",
                                   "softmax_scale": "entropy_invariance", "train_avg_length": 12185, "use_pe_proj": True}]}


def build_head(device, shipped=False, conv_precision=None):
    from mvsformerplusplus_amd import synth
    from mvsformerplusplus_amd.cascade import CascadeDepthHead
    args = json.loads(json.dumps(dict(ARGS, **SHIPPED))) if shipped else dict(ARGS)
    if conv_precision:
        args["conv_precision"] = conv_precision
    head = CascadeDepthHead(args)
    for i, st in enumerate(head.fusions):
        st.load_state_dict(synth.seeded_state_dict(synth.state_dict_manifest(st.state_dict()), 100 + i), strict=True)
        st.return_prob_volumes = True       # reference-faithful outputs (a12): prob_volume / prob_volume_pre are written
    return head.eval().to(device)


class HeadlineGuard:
    """Keeps rank 0's ONE JSON line alive across a leg that may take the process down (first RCCL contact of the view-sharded mode): a
    small detached Python process (own session, SIGTERM / SIGINT / SIGHUP ignored, this process's stdout) reads the fallback line from
    a pipe and prints it if the pipe closes before `disarm()` said that this process will print the line itself."""
    _CHILD = ("import sys, signal\n"
              "for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP): signal.signal(s, signal.SIG_IGN)\n"
              "line = sys.stdin.readline()\n"
              "ok = sys.stdin.readline()\n"
              "if ok.strip() != 'OK':\n"
              "    sys.stdout.write(line if line.endswith('\\n') else line + '\\n'); sys.stdout.flush()\n")

    def __init__(self, fallback: dict):
        import subprocess
        self.proc = None
        try:                                   # the guard is a convenience: if the helper cannot be started the leg runs unguarded
            sys.stdout.flush()
            self.proc = subprocess.Popen([sys.executable, "-c", self._CHILD], stdin=subprocess.PIPE, start_new_session=True, close_fds=True)
            self.proc.stdin.write((json.dumps(fallback) + "\n").encode())
            self.proc.stdin.flush()
        except Exception as e:
            print("bench.py: headline guard not started (%r)" % (e,), file=sys.stderr)
            self.proc = None

    def disarm(self):
        if self.proc is None:
            return
        try:
            self.proc.stdin.write(b"OK\n")
            self.proc.stdin.close()
            self.proc.wait(timeout=10)
        except Exception:
            pass
        self.proc = None


def cpu_baseline(head, feats, projs, dv, max_threads=None, shipped=False):
    """The oracle (CPU restatement of the reference PyTorch path, fp32) on the host cores, SURVEY.md section 8d: one warm-up pass
    (the first call is slower), then the MEDIAN of three passes on 16 threads, and the median of three on 8 threads beside it (the
    thread count of the in-container reference measurements)."""
    from oracle import ref_path as O
    # oneDNN convolutions on these shapes get SLOWER beyond a few dozen threads (256 threads: 171 s per pass on the
    # GPU box vs 11.5 s on 8 threads in the build container), so the baseline uses 16 threads and says so.
    sds = [{k: v.detach().cpu() for k, v in st.state_dict().items()} for st in head.fusions]
    f = {k: v.float().cpu() for k, v in feats.items()}
    p = {k: v.cpu() for k, v in projs.items()}
    d = dv.cpu()

    def passes(nthreads, count):
        torch.set_num_threads(nthreads)
        ts, o = [], None
        with torch.no_grad():
            for _ in range(count):
                t0 = time.time()
                o = O.cascade_forward(f, p, d, sds, ndepths=ARGS["ndepths"], depth_interals_ratio=ARGS["depth_interals_ratio"],
                                      base_ch=ARGS["base_ch"], tmp=TMP, use_pe3d=shipped,
                                      transformer_config=SHIPPED["transformer_config"] if shipped else None)
                ts.append(time.time() - t0)
        return ts, o

    n = min(max_threads or 16, os.cpu_count() or 1)
    t_main, out = passes(n, 4)                                   # pass 0 = warm-up
    med = sorted(t_main[1:])[1]
    res = {"value": 1.0 / med, "unit": "ref-views/s", "cores": n, "kind": "port",
           "sample": "1 reference view of the bench workload (full 4-stage cascade, fp32) per pass; warm-up pass %.1f s, then the median of "
                     "three passes: %s s" % (t_main[0], " / ".join("%.1f" % t for t in t_main[1:]))}
    if n > 8:
        t8, _ = passes(8, 3)
        res["at_8_threads"] = {"value": 1.0 / sorted(t8)[1], "cores": 8, "passes_s": [round(t, 2) for t in t8]}
    return res, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views-per-step", type=int, default=96,
                    help="reference views in one step (= one batch of synthetic input); 20 steps x 96 views keep the timed region >= 2 s up to 960 ref-views/s")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="distinct synthetic input sets the reference views rotate over (4 x 531 MB of features > the 256 MiB Infinity "
                         "Cache: no step finds its inputs cache-resident by construction)")
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1536)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1,
                    help="reference views per forward call (the module API's batch axis B).  1 = the reference's own loop (test.py feeds one "
                         "reference view per call); larger batches put B views into every launch (B x the workgroups per launch, 1 / B of the "
                         "launches per view)")
    ap.add_argument("--issue", choices=["graph", "eager"], default="graph",
                    help="graph (default): replay one captured hipGraph per (stream, input set) (CascadeDepthHead.capture) - one launch per "
                         "reference view from the host's side; eager: issue the ~60 launches of a reference view from Python.  The same device work")
    ap.add_argument("--graph", action="store_true", help="= --issue graph (kept for the round-3/4 scripts)")
    ap.add_argument("--no-shipped-leg", action="store_true", help="skip the extra `shipped` object (stage-1 transformer mix: value + parity)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-composite-baseline", action="store_true", help="skip the torch-ROCm composite baseline leg (the oracle's tensors on cuda:0)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the extra `training_step` object (forward + backward of each cascade stage)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-table", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the reference views of a step are issued on round-robin (independent views overlap: the small "
                         "coarse-stage launches of one view run beside the large fine-stage launches of another, SURVEY.md section 8e); "
                         "the single-stream latency is reported beside it")
    ap.add_argument("--feat-dtype", choices=["fp32", "bf16", "fp16"], default="fp32",
                    help="dtype of the feature maps handed to the path (the reference's FPN emits bf16 under test.py:250's autocast and "
                         "StageNet upcasts per view, cost_volume.py:67); the headline keeps fp32")
    ap.add_argument("--feat-layout", choices=["planar", "tiled", "emitted"], default="planar",
                    help="planar: [B,V,C,H,W] as the reference's FPN emits it (headline); tiled: the octet-tiled channel-last hand-off "
                         "layout [B,V,C/8,H,W,8] (SURVEY.md section 8f #4) packed once by mvs_pack_features, outside the timed region; "
                         "emitted: stages 2-4 come out of the producer-side emitter itself (TiledFeatureHead = mvs_conv2d3x3_tiles_fwd, the feature "
                         "side's last 3x3 convolution writing bf16 octet tiles from its epilogue, FMT.py:195-197) applied to the synthetic maps - "
                         "mvs_pack_features never runs in the process; stage 1 (the FMT's own output in the reference, no convolution) stays planar")
    ap.add_argument("--emit-dtype", choices=["bf16", "fp16"], default="bf16",
                    help="--feat-layout emitted: dtype of the emitter's octet tiles.  fp16 tiles are what the fp16 gather forms hold in LDS anyway "
                         "(same values as fp32 features rounded once); with them the fine stages' gather needs no window at all - a tap of 8 channels "
                         "is one 16-byte run, four buffer loads per plane straight from HBM / L2 (gather_lds.h, MVS_GL_DIRECT16)")
    ap.add_argument("--view-sharded-timeout", type=int, default=120, help="N > 1: seconds the extra view-sharded latency leg may take")
    ap.add_argument("--view-sharded-only", action="store_true",
                    help="N > 1: run ONLY the view-sharded latency mode (SURVEY.md section 8e: the source views of ONE reference view over the ranks, "
                         "RCCL all-reduce / slab exchange per stage, BASELINE configs[2]'s V = 10) and print its JSON line: value = reference views "
                         "per second of the whole group, scaling 'strong'.  For a first RCCL contact without the data-parallel headline in front of it")
    ap.add_argument("--conv-precision", choices=["stagemix", "auto", "bf16x3", "f16x2", "f16mix", "f16", "fp32"], default=None,
                    help="contraction / activation format of the 3-D regularisers (default: the package default, cost_volume.STAGE_DEFAULT_PRECISION)")
    ap.add_argument("--keep-min-depth", type=int, default=None,
                    help="A/B: cost_volume.KEEP_MIN_DEPTH (planes from which the fp16 gather form keeps fp16 correlations; 1 = stage 4's D = 4 too)")
    ap.add_argument("--keep-exact-min-depth", type=int, default=None,
                    help="A/B: cost_volume.KEEP_EXACT_MIN_DEPTH (planes from which the exact coarse-stage gather keeps fp32 correlations instead of gathering twice)")
    ap.add_argument("--no-keep-correlations", action="store_true",
                    help="A/B: pass 2 of every stage gathers again instead of streaming the fp16 correlations kept by pass 1 (StageNet.keep_correlations)")
    ap.add_argument("--cost-reg", choices=["normal", "shipped"], default="normal",
                    help="normal: all-'Normal' regularisers (Track R headline); shipped: stage-1 transformer + PE3D as in the shipped config")
    a = ap.parse_args()
    a.graph = a.graph or a.issue == "graph"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            print("bench.py --gpus %d must be launched with torch.distributed.run (see docstring)" % a.gpus, file=sys.stderr)
            sys.exit(2)
    import torch.distributed as dist
    # test hooks (scripts/gpu_round.sh runs the N = 2 flow on a one-GPU box with them): all ranks on one device, gloo instead of RCCL
    if os.environ.get("MVS_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("MVS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from mvsformerplusplus_amd import profiling, synth
    if a.keep_exact_min_depth is not None or a.keep_min_depth is not None:
        from mvsformerplusplus_amd import cost_volume as _cvm
        if a.keep_exact_min_depth is not None:
            _cvm.KEEP_EXACT_MIN_DEPTH = a.keep_exact_min_depth
        if a.keep_min_depth is not None:
            _cvm.KEEP_MIN_DEPTH = a.keep_min_depth
    head = build_head(device, shipped=a.cost_reg == "shipped", conv_precision=a.conv_precision)
    if a.no_keep_correlations:
        for st in head.fusions:
            st.keep_correlations = False
    fdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[a.feat_dtype]
    if a.view_sharded_only:
        if world < 2:
            print("--view-sharded-only needs N > 1 ranks (torch.distributed.run)", file=sys.stderr)
            sys.exit(2)

        def _sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        vs = view_sharded_leg(head, a, device, world, rank, _sync, fdt)
        if rank == 0:
            print(json.dumps({"metric": "ref-views/sec at 1152x1536 N=5 D=192 4-stage; achieved HBM GB/s vs peak", "mode": "view-sharded-only",
                              "value": vs["ref_views_per_s"], "unit": "ref-views/s", "n_gpus": world, "steps": max(3, a.steps // 4), "warmup": 2,
                              "ms_per_step": vs["ms_per_ref_view"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": DTYPE_TEXT[head.fusions[0].precision_policy], "data": "synthetic",
                              "config": {"workload": "BASELINE configs[2]: 1152x1536, V = %d, 4-stage cascade, source views sharded over %d ranks" % (vs["views"], world),
                                         "parallelism": "source views over %d ranks, one reference view at a time" % world},
                              "view_sharded": vs}))
        dist.barrier()
        dist.destroy_process_group()
        return
    nsets = max(1, a.input_sets)
    BATCH = max(1, a.batch)
    sets = [synth.make_cascade_inputs(a.height, a.width, a.views, seed=100 * rank + i, device=device, feat_dtype=fdt, batch=BATCH) for i in range(nsets)]
    if a.feat_layout == "tiled":
        from mvsformerplusplus_amd import ops
        sets = [({k: ops.pack_features(v) for k, v in f.items()}, p, d) for f, p, d in sets]
    elif a.feat_layout == "emitted":
        import torch.nn as nn
        from mvsformerplusplus_amd import TiledFeatureHead
        heads_e = {}
        for k, C in (("stage2", 32), ("stage3", 16), ("stage4", 8)):
            conv = nn.Conv2d(C, C, 3, padding=1, bias=False)
            with torch.no_grad():                             # identity + a small random 3x3 part: the views stay correlated, the convolution is real work
                conv.weight.mul_(0.1)
                conv.weight[torch.arange(C), torch.arange(C), 1, 1] += 1.0
            heads_e[k] = TiledFeatureHead(conv, dtype=torch.float16 if a.emit_dtype == "fp16" else torch.bfloat16).to(device)
        sets = [({k: (heads_e[k](v.float()) if k in heads_e else v) for k, v in f.items()}, p, d) for f, p, d in sets]
    feats, projs, dv = sets[0]
    R = max(1, a.views_per_step)
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    out = None
    with torch.no_grad():
        # untimed pre-warm: code objects, allocator pools and the clock governor of a fresh box settle before the W warm-up steps
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.3:
            out = head(feats, projs, dv, tmp=TMP)
            torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=device) for _ in range(a.streams)] if a.streams > 1 else None

        def make_runner(hd, sets=sets):
            """-> (run(n_steps, first), issue text).  Graph mode: one captured hipGraph per (stream, input set); a capture that fails
            falls back to eager issue and says so in config.issue (the headline must not die on it)."""
            graphs, issue = {}, "eager launches"
            if a.graph:
                try:
                    for si in range(len(streams) if streams is not None else 1):
                        for k in range(nsets):
                            if streams is None:
                                graphs[(si, k)] = hd.capture(*sets[k], tmp=TMP)
                            else:
                                with torch.cuda.stream(streams[si]):
                                    graphs[(si, k)] = hd.capture(*sets[k], tmp=TMP)
                    torch.cuda.synchronize()
                    issue = "one hipGraph replay per reference view"
                except Exception as e:
                    graphs, issue = {}, "eager launches (hipGraph capture failed: %r)" % (e,)
                    torch.cuda.synchronize()

            def run(n_steps, first=0):
                """n_steps steps; step k = the R reference views k*R .. k*R+R-1, view j on input set j % nsets and stream j % nstreams."""
                o = None
                if streams is not None:
                    for st in streams:
                        st.wait_stream(torch.cuda.current_stream(device))
                for j in range(first * R // BATCH, (first + n_steps) * R // BATCH):       # one forward call = BATCH reference views
                    f, p, d = sets[j % nsets]
                    if streams is None:
                        o = graphs[(0, j % nsets)]() if graphs else hd(f, p, d, tmp=TMP)
                    else:
                        with torch.cuda.stream(streams[j % len(streams)]):
                            o = graphs[(j % len(streams), j % nsets)]() if graphs else hd(f, p, d, tmp=TMP)
                if streams is not None:
                    for st in streams:
                        torch.cuda.current_stream(device).wait_stream(st)
                return o
            return run, issue

        run, issue_text = make_runner(head)

        run(a.warmup)
        sync_all()
        t0 = time.perf_counter()
        run(a.steps, first=a.warmup)
        sync_all()
        elapsed = time.perf_counter() - t0

        # single-stream latency: one reference view at a time on the current stream, the reference's own loop shape (test.py:238-252)
        n_lat = 4 * nsets
        for j in range(nsets):
            head(*sets[j], tmp=TMP)
        torch.cuda.synchronize()
        tl0 = time.perf_counter()
        for j in range(n_lat):
            head(*sets[j % nsets], tmp=TMP)
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - tl0) / (n_lat * BATCH) * 1e3
        out = head(feats, projs, dv, tmp=TMP)
        torch.cuda.synchronize()
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3
    value = world * a.steps * R / elapsed
    is_cfg2 = (a.height, a.width, a.views) == (1152, 1536, 5)

    result = {
        "metric": "ref-views/sec at 1152x1536 N=5 D=192 4-stage; achieved HBM GB/s vs peak",
        "value": value, "unit": "ref-views/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",           # replaced below by the format the regularisers really compute in
        "config": {"workload": WORKLOAD if is_cfg2 else "%dx%d V=%d 4-stage cascade" % (a.height, a.width, a.views),
                   "height": a.height, "width": a.width, "views": a.views, "global_batch": world * R,
                   "step": "one batch of %d reference views (batch %d per forward call%s), rotating over %d input sets" %
                           (R, BATCH, ", like test.py" if BATCH == 1 else "", nsets),
                   "views_per_forward_call": BATCH, "issue": issue_text,
                   "ref_views_per_step_per_gpu": R, "input_sets": nsets, "timed_seconds": elapsed,
                   "parallelism": "dp%d over reference views" % world, "streams_per_gpu": a.streams,
                   "features": "%s %s resident in HBM" % (a.feat_dtype, {"tiled": "octet-tiled [B,V,C/8,H,W,8]", "planar": "planar [B,V,C,H,W]",
                                                                           "emitted": "planar stage 1 + %s octet tiles written by the producer-side emitter (stages 2-4)" % a.emit_dtype}[a.feat_layout])},
        "ms_per_ref_view": ms_per_step / R,
        "latency": {"single_stream_ms_per_ref_view": latency_ms, "single_stream_ref_views_per_s": 1e3 / latency_ms,
                    "note": "one reference view at a time on one stream (the reference's loop, test.py:238-252), same rotating inputs"},
    }
    if is_cfg2:
        gbs = ALGO_BYTES_PER_VIEW * value / world / 1e9
        result["whole_path"] = {"algorithmic_bytes_per_ref_view": ALGO_BYTES_PER_VIEW, "achieved_gbs_per_gpu": gbs, "peak_gbs": 8000.0,
                                "frac": gbs / 8000.0, "frac_single_stream": ALGO_BYTES_PER_VIEW / (latency_ms * 1e-3) / 8.0e12,
                                "note": "frac: SURVEY.md section 8d layer-wise byte model (fp32 tensors, 5.03 GB per reference view at cfg2) x ref-views/s / "
                                        "8 TB/s; frac_as_built / frac_pmc (added by the profile leg): the bytes the tensors of the as-built format "
                                        "really have / the PMC-counted HBM bytes, same views/s, same peak"}

    result["config"]["cost_reg_type"] = SHIPPED["cost_reg_type"] if a.cost_reg == "shipped" else ["Normal"] * 4
    prec0 = head.fusions[0].precision_policy
    result["config"]["conv_precision"] = {
        "bf16x3": "bf16x3 MFMA contraction, fp32 accumulation; fp32-equivalent activations (between the U-Net layers stored as split hi | lo bf16 pairs, "
                  "the same 4 bytes per element)",
        "f16x2": "f16x2: U-Net activations (cost volume included) stored as fp16, weights as fp16 hi + lo, two MFMA terms per product on "
                 "v_mfma_f32_16x16x32_f16, fp32 accumulation; warp / correlation / visibility / heads in fp32 (the reference's GPU path runs the "
                 "regulariser under bf16 autocast, test.py:250)",
        "f16mix": "f16mix (every stage): fp16 U-Net activations as in f16x2; weights fp16 hi + lo (two MFMA terms) on the 8- / 16-channel layers, "
                  "ONE fp16 term on the 32- / 64-channel layers conv4..conv7 (no measurable change of the depth error, scripts/study_weight_precision.py); "
                  "fp32 accumulation; warp / correlation / visibility / heads in fp32 (the reference's GPU path runs the regulariser under bf16 autocast, "
                  "test.py:250)",
        "f16": "f16: fp16 U-Net activations, ONE fp16 weight term on every layer (depth 7e-5 / 4.8e-4 from the fp32 oracle on plain / stress inputs)",
        "stagemix": "stagemix (product default policy, round 5): the coarse stages (ndepth > model_th: CostRegNet, D = 32 / 16 - their depth schedules the "
                    "next stage's hypotheses) run fp32-equivalent: split-bf16 (bf16x3, three MFMA terms) regulariser and visibility CNN, exact gather (fp32 "
                    "source windows, fp32 kept correlations); the CostRegNet3D stages (D = 8 / 4) run f16mix (fp16 activations; fp16 hi + lo weights on "
                    "the 8- / 16-channel layers, one fp16 term on conv4..conv7 and in the visibility CNN; fp16 source windows and kept correlations); fp32 "
                    "accumulation everywhere (the reference's GPU path runs the regulariser under bf16 autocast, test.py:250)",
        "fp32": "fp32-exact MFMA contraction"}[prec0]
    result["dtype"] = DTYPE_TEXT[prec0]
    if getattr(head, "_auto", False):
        result["config"]["conv_precision"] = ("auto (the cascade's default policy since round 6: the uniform fp16 format while depth_max / depth_min stays below half the ratio at which the "
                                              "inverse-depth schedule degenerates, the exact coarse stages of 'stagemix' otherwise): chose '%s' for this workload's depth range "
                                              "(depth_max / depth_min = %.2f against the critical 12.6); " % (prec0, float(dv.max() / dv.min()))
                                              + result["config"]["conv_precision"])
        result["config"]["precision_policy"] = "auto -> " + prec0
    if prec0 in ("f16x2", "f16mix", "f16", "stagemix"):
        result["config"]["gather_pass2"] = ("second gather on every stage (--no-keep-correlations)" if a.no_keep_correlations else
                                            "stream of the per-view correlations kept by pass 1 (fp32 on the exact coarse stages from D = 16 on, fp16 "
                                            "on the fp16-format stages with D > 4); otherwise a second gather")

    # ---- per-kernel HIP-event profile -> roofline of the dominant kernel ----
    if not a.no_profile:
        reps = 4
        allruns = []
        for r in range(reps):
            _, launches = profiling.profile_cascade(head, *sets[r % nsets], TMP)
            allruns.append(launches)
        agg = profiling.summarize([l for run in allruns for l in run])
        # the dominant KERNEL = the __global__ function with the largest share of a reference view's kernel time, ALL of its template
        # instantiations together (round 5, VERDICT r4 weak #4: per-instantiation ranking made the visibility CNN "dominant" at 8 % of the step
        # only because the convolutions are spread over ~25 tile configurations).  Bundles of several launches timed as a unit are not candidates.
        groups = {}
        for k, v in agg.items():
            if k.startswith("[bundle]"):
                continue
            gname = profiling.kernel_group(k)
            gr = groups.setdefault(gname, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "issued_flops": 0.0, "members": []})
            for f in ("calls", "ms", "flops", "bytes", "issued_flops"):
                gr[f] += v[f]
            gr["members"].append(k)
        dom_name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        dom["avg_ms"] = dom["ms"] / dom["calls"]
        dom["tflops"] = dom["flops"] / max(dom["ms"], 1e-9) / 1e9
        dom["gbs"] = dom["bytes"] / max(dom["ms"], 1e-9) / 1e6
        mfma = dom_name.startswith("conv3d") or dom_name.startswith("deconv3d") or dom_name.startswith("vis_cnn")
        per_launch = dom["flops" if mfma else "bytes"] / dom["calls"]
        achieved = (dom["tflops"] if mfma else dom["gbs"])
        prec = head.fusions[0].precision_policy          # "stagemix" (default): bf16x3 on the coarse stages, f16mix on the fine ones
        terms = dom["issued_flops"] / max(dom["flops"], 1e-12) if mfma else 1      # MFMA products issued per algorithmic product, launch-weighted
        if not mfma:
            peak, note = profiling.PEAK_HBM_GBS, ("algorithmic HBM bytes per launch (SURVEY.md section 8d: every feature map once + hypotheses once + "
                                                  "outputs once, at the tensors' real element sizes) / HIP-event launch time on the launch stream")
        elif prec != "fp32":
            # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak 2.5 PFLOP/s.  `frac` is against THAT peak (the guide's); the contraction issues
            # `terms` MFMA products per algorithmic product, an implementation choice reported separately as frac_of_issued_mfma
            peak, note = profiling.PEAK_F16_MFMA_TFLOPS, ("algorithmic FLOPs (2 x MACs of the operator) of ALL launches of this __global__ function in one reference view "
                                                         "(every tile configuration / cascade stage) / their summed HIP-event time vs the dense fp16 / bf16 MFMA peak of "
                                                         "MI355X_MICROARCH.md; the launches issue %.2f MFMA term(s) per algorithmic product on average" % terms)
        else:
            peak, note = profiling.PEAK_F32_MFMA_TFLOPS, "fp32-exact contraction on v_mfma_f32_16x16x4_f32 (157.3 TFLOP/s dense peak)"
        traffic, tsrc = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # PMC-derived HBM bytes per launch, committed per round
        if os.path.exists(tfile):
            traffic, tsrc = profiling.group_pmc_traffic(dom_name, json.load(open(tfile)))
        result["roofline"] = {"kernel": dom_name, "instantiations": sorted(dom["members"]), "bound": "mfma" if mfma else "hbm", "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s" if mfma else "GB/s", "frac": achieved / peak, "traffic": traffic, "traffic_pmc_symbols": tsrc,
                              "avg_launch_ms": dom["avg_ms"], "algorithmic_per_launch": per_launch,
                              "share_of_step": dom["ms"] / max(sum(v["ms"] for v in agg.values()), 1e-9),
                              "launches_per_ref_view": dom["calls"] / reps, "note": note}
        if mfma:
            result["roofline"]["mfma_terms_per_product"] = terms
            result["roofline"]["frac_of_issued_mfma"] = achieved * terms / peak
        if mfma:
            result["roofline"]["algorithmic_bytes_per_launch"] = dom["bytes"] / dom["calls"]
            result["roofline"]["hbm_view"] = {"achieved_gbs": dom["gbs"], "frac_of_8TBs": dom["gbs"] / profiling.PEAK_HBM_GBS,
                                              "note": "the same launches by their as-built algorithmic bytes: the family is bound by neither roof (DESIGN.md 4.2)"}
        # the two gather passes (the kernels VERDICT r1 named: 6 % of the HBM roofline then), per instantiation, by the SURVEY 8d byte count
        result["gather_roofline"] = {k: {"achieved_gbs": v["gbs"], "frac_of_8TBs": v["gbs"] / profiling.PEAK_HBM_GBS, "avg_launch_ms": v["avg_ms"],
                                         "algorithmic_bytes_per_launch": v["bytes"] / v["calls"]}
                                     for k, v in agg.items() if k.startswith(("gl_", "warp_corr_", "corr_aggregate"))}
        gb = sum(v["bytes"] for k, v in agg.items() if k.startswith(("gl_", "warp_corr_", "corr_aggregate")))
        gt = sum(v["ms"] for k, v in agg.items() if k.startswith(("gl_", "warp_corr_", "corr_aggregate")))
        result["gather_roofline"]["all_passes"] = {"achieved_gbs": gb / max(gt, 1e-9) / 1e6, "frac_of_8TBs": gb / max(gt, 1e-9) / 1e6 / profiling.PEAK_HBM_GBS,
                                                   "ms_per_ref_view": gt / reps}
        result["kernels"] = {k: {"calls_per_ref_view": v["calls"] / reps, "ms_per_ref_view": v["ms"] / reps, "gbs": v["gbs"], "tflops": v["tflops"]}
                             for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        # kernel FAMILIES: time, algorithmic work by SURVEY.md section 8d and the fraction of the family's own peak, so
        # that the roofline picture does not hinge on which single symbol happens to be the largest
        def fam(pred, bound, fam_prec=None):
            ks = {k: v for k, v in agg.items() if pred(k)}
            ms = sum(v["ms"] for v in ks.values()) / reps
            if not ks or ms <= 0:
                return None
            if bound == "mfma":
                work = sum(v["flops"] for v in ks.values()) / reps
                ach = work / ms / 1e9                            # TFLOP/s
                fp = fam_prec or prec
                issued = sum(v["issued_flops"] for v in ks.values()) / reps / ms / 1e9                       # TFLOP/s of MFMA products actually issued
                pk = profiling.PEAK_F32_MFMA_TFLOPS if fp == "fp32" else profiling.PEAK_F16_MFMA_TFLOPS
                return {"ms_per_ref_view": ms, "bound": "mfma", "algorithmic_gflop_per_ref_view": work / 1e9, "achieved_tflops": ach, "peak_tflops": pk,
                        "frac": ach / pk, "mfma_terms_per_product": issued / max(ach, 1e-12), "frac_of_issued_mfma": issued / pk,
                        "launches_per_ref_view": sum(v["calls"] for v in ks.values()) / reps}
            work = sum(v["bytes"] for v in ks.values()) / reps
            ach = work / ms / 1e6                                # GB/s
            return {"ms_per_ref_view": ms, "bound": "hbm", "algorithmic_mb_per_ref_view": work / 1e6, "achieved_gbs": ach, "peak_gbs": profiling.PEAK_HBM_GBS,
                    "frac": ach / profiling.PEAK_HBM_GBS, "launches_per_ref_view": sum(v["calls"] for v in ks.values()) / reps}
        is_gather = lambda k: k.startswith(("gl_", "warp_corr_", "corr_aggregate"))
        is_conv = lambda k: k.startswith(("conv3d_mfma", "deconv3d_mfma"))
        is_vis = lambda k: k.startswith(("vis_",)) or k.startswith("conv3d_mfma<16,16,k1") or k.startswith("conv3d_mfma<16,8,k1")
        families = {"gather": fam(is_gather, "hbm"), "visibility_cnn": fam(is_vis, "mfma"),
                    "regulariser_convolutions": fam(lambda k: is_conv(k) and not is_vis(k), "mfma"),
                    "heads_and_ranges": fam(lambda k: not (is_gather(k) or is_conv(k) or is_vis(k) or k.startswith(("tr_", "[bundle]"))), "hbm")}
        if families["gather"] is not None and is_cfg2:
            # the gather phase by SURVEY 8d's PATH count (features once, hypotheses once, volume once, entropy + visibility maps: 1.08 GB per
            # reference view) beside the per-launch count above (features once PER PASS)
            g = families["gather"]
            g["by_path_count"] = {"algorithmic_mb_per_ref_view": 1080.0, "achieved_gbs": 1080.0 / g["ms_per_ref_view"], "frac": 1080.0 / g["ms_per_ref_view"] / profiling.PEAK_HBM_GBS}
        if families["regulariser_convolutions"] is not None:
            c = families["regulariser_convolutions"]
            cb = sum(v["bytes"] for k, v in agg.items() if is_conv(k) and not is_vis(k)) / reps
            c["hbm_view"] = {"algorithmic_mb_per_ref_view": cb / 1e6, "achieved_gbs": cb / c["ms_per_ref_view"] / 1e6,
                             "frac": cb / c["ms_per_ref_view"] / 1e6 / profiling.PEAK_HBM_GBS}
        result["families"] = {k: v for k, v in families.items() if v is not None}
        if "whole_path" in result:
            vps = value / world
            ab = sum(v["bytes"] for k, v in agg.items() if not k.startswith("[bundle]")) / reps
            wp = result["whole_path"]
            wp["as_built_bytes_per_ref_view"] = ab
            wp["frac_as_built"] = ab * vps / 8.0e12
            wp["as_built_note"] = ("sum over the launches of one reference view of inputs read once + outputs written once at the tensors' real element "
                                   "sizes (fp16 regulariser activations: 2 bytes; features counted once per gather pass)")
            if os.path.exists(tfile):
                tj = json.load(open(tfile))
                pm, miss = 0.0, []
                for k, v in agg.items():
                    if k.startswith("[bundle]"):
                        continue
                    hit = profiling.match_kernel(k, tj)
                    if hit is None or tj[hit].get("hbm_bytes_per_launch") is None:
                        miss.append(k)
                    else:
                        pm += tj[hit]["hbm_bytes_per_launch"] * v["calls"] / reps
                if "_whole_path" in tj and a.cost_reg != "shipped" and a.conv_precision is None:      # the PMC run is the default policy's
                    pm, miss = tj["_whole_path"]["hbm_bytes_per_ref_view"], []       # every launch of the PMC run / its reference views
                wp["pmc_bytes_per_ref_view"] = pm
                wp["frac_pmc"] = pm * vps / 8.0e12
                wp["pmc_symbols_without_counters"] = miss
                wp["pmc_note"] = ("profiles/pmc_traffic.json: HBM-side bytes counted by rocprofv3 PMC passes of this bench command in the product default format "
                                  "(committed per round, not re-measured by this run)")
        if a.profile_table and rank == 0:
            tot = sum(v["ms"] for v in agg.values()) / reps
            print("%-40s %6s %9s %9s %9s" % ("kernel", "calls", "ms/view", "GB/s", "TFLOP/s"), file=sys.stderr)
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
                print("%-40s %6.0f %9.3f %9.0f %9.1f" % (k, v["calls"] / reps, v["ms"] / reps, v["gbs"], v["tflops"]), file=sys.stderr)
            print("sum of kernel times %.3f ms/ref view; wall %.3f ms/ref view (%d streams), %.3f single stream" %
                  (tot, ms_per_step / R, a.streams, latency_ms), file=sys.stderr)

    # ---- shipped mix with bf16 attention probabilities (optional fast mode of the transformer stage), N = 1 ----
    if world == 1 and a.cost_reg == "shipped" and not a.no_profile:
        try:
            head.fusions[0].cost_reg.attention_precision = "bf16p"
            with torch.no_grad():
                for _ in range(2):
                    o2 = head(feats, projs, dv, tmp=TMP)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    o2 = head(feats, projs, dv, tmp=TMP)
                torch.cuda.synchronize()
                tb = (time.perf_counter() - t0) / 5
            d1, d2 = out["refined_depth"], o2["refined_depth"]
            result["attention_bf16p"] = {"value": 1.0 / tb, "unit": "ref-views/s", "ms_per_ref_view": tb * 1e3,
                                         "refined_depth_rel_l1_vs_default": float(((d2 - d1).abs() / d1.abs()).mean()),
                                         "note": "attention_precision='bf16p': softmax probabilities enter p.v as one bf16 term"}
            head.fusions[0].cost_reg.attention_precision = "bf16x3"
        except Exception as e:
            result["attention_bf16p"] = {"error": repr(e)}

    # ---- extra: the fp32-equivalent regulariser format ("bf16x3") on the same weights, inputs and loop, outside the timed headline: the
    #      driver's own BENCH line then carries both modes (the headline runs the product default policy) ----
    def side_leg(hd, n2, use_sets=None):
        """ref-views/s of another head on the headline's inputs (or `use_sets`), streams and issue mode (n2 timed steps after one warm-up step)
        + its outputs on input set 0."""
        with torch.no_grad():
            run2, issue2 = make_runner(hd) if use_sets is None else make_runner(hd, use_sets)
            run2(1)
            sync_all()
            t0 = time.perf_counter()
            run2(n2, first=1)
            sync_all()
            dt = (time.perf_counter() - t0) / n2
            o = hd(feats, projs, dv, tmp=TMP) if use_sets is None else hd(*use_sets[0], tmp=TMP)
            o = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}
            torch.cuda.synchronize()
        del run2
        return dt, o, issue2

    if world == 1 and not a.no_profile and a.cost_reg != "shipped" and head.fusions[0].precision_policy in ("f16x2", "f16mix", "f16", "stagemix"):
        try:
            head32 = build_head(device, conv_precision="bf16x3")
            n2 = max(2, a.steps // 4)
            t32, out32, _ = side_leg(head32, n2)
            d16, d32 = out["refined_depth"], out32["refined_depth"]
            result["fp32_equivalent_mode"] = {"conv_precision": "bf16x3", "value": R / t32, "unit": "ref-views/s", "ms_per_ref_view": t32 / R * 1e3, "steps": n2,
                                              "default_vs_this_refined_depth_rel_l1": float(((d16 - d32).abs() / d32.abs()).mean()),
                                              "note": "split-bf16 activations (hi | lo pairs, 4 bytes per element), three MFMA terms on every stage: 1e-6 from the fp32 oracle"}
            del head32, out32
            torch.cuda.empty_cache()
        except Exception as e:
            result["fp32_equivalent_mode"] = {"error": repr(e)}
        # ... and, when the default policy ("auto") chose the uniform fp16 format for this depth range, what its other branch - the exact coarse
        # stages of "stagemix", which it takes on ill-conditioned ranges such as BASELINE cfg4 / cfg5's - costs on the same inputs
        if getattr(head, "_auto", False) and head.fusions[0].precision_policy != "stagemix":
            try:
                headx = build_head(device, conv_precision="stagemix")
                n2 = max(2, a.steps // 4)
                tx, outx, _ = side_leg(headx, n2)
                result["exact_coarse_mode"] = {"conv_precision": "stagemix", "value": R / tx, "unit": "ref-views/s", "ms_per_ref_view": tx / R * 1e3, "steps": n2,
                                               "default_vs_this_refined_depth_rel_l1": float(((out["refined_depth"] - outx["refined_depth"]).abs() / outx["refined_depth"].abs()).mean()),
                                               "note": "coarse stages (D = 32 / 16) fp32-equivalent: split-bf16 regulariser and visibility CNN, exact gather; fine stages f16mix - the branch the "
                                                       "default policy takes when depth_max / depth_min exceeds half the critical ratio (round 5's unconditional default)"}
                del headx, outx
                torch.cuda.empty_cache()
            except Exception as e:
                result["exact_coarse_mode"] = {"error": repr(e)}
        # ... and rounds 3-4's default, the uniform fp16 format on every stage (it leaves the 1e-3 bar on ill-conditioned
        # depth ranges) - what the exact coarse stages cost on this workload, when they are what the headline ran
        if head.fusions[0].precision_policy == "stagemix":
            try:
                head16 = build_head(device, conv_precision="f16mix")
                n2 = max(2, a.steps // 4)
                t16, out16, _ = side_leg(head16, n2)
                result["uniform_f16mix_mode"] = {"conv_precision": "f16mix", "value": R / t16, "unit": "ref-views/s", "ms_per_ref_view": t16 / R * 1e3, "steps": n2,
                                                 "default_vs_this_refined_depth_rel_l1": float(((out["refined_depth"] - out16["refined_depth"]).abs() / out16["refined_depth"].abs()).mean()),
                                                 "note": "fp16 U-Net tensors, source windows and kept correlations on EVERY stage (rounds 3-4's default): 4.7e-5 from the fp32 oracle at cfg2, "
                                                         "4e-3 on BASELINE cfg4 / cfg5's literal depth range - opt-in (conv_precision='f16mix')"}
                del head16, out16
                torch.cuda.empty_cache()
            except Exception as e:
                result["uniform_f16mix_mode"] = {"error": repr(e)}
        # ... and the default policy fed with the hand-off a producer-side emitter gives it: the SAME features of
        # stages 3-4 (the fp16 gather forms) as fp16 octet tiles, where a tap of 8 channels is one 16-byte run and the gather needs no LDS
        # window (gather_lds.h, MVS_GL_DIRECT16); packed here by mvs_pack_features outside the timed region, like every input
        if a.feat_layout == "planar" and a.feat_dtype == "fp32":
            try:
                from mvsformerplusplus_amd import ops as _ops
                sets_t = [({k: (_ops.pack_features(v, torch.float16) if k in ("stage3", "stage4") else v) for k, v in f.items()}, p, d) for f, p, d in sets]
                n2 = max(2, a.steps // 4)
                tt, outt, _ = side_leg(head, n2, use_sets=sets_t)
                result["fp16_tiles_handoff_mode"] = {"features": "stages 3-4 as fp16 octet tiles [B,V,C/8,H,W,8] (TiledFeatureHead / mvs_conv2d3x3_tiles_fwd with fp16 output, or "
                                                                 "mvs_pack_features), stages 1-2 as in the headline", "value": R / tt, "unit": "ref-views/s",
                                                     "ms_per_ref_view": tt / R * 1e3, "steps": n2,
                                                     "default_vs_this_refined_depth_rel_l1": float(((out["refined_depth"] - outt["refined_depth"]).abs() / outt["refined_depth"].abs()).mean()),
                                                     "note": "same source-window values as the headline (its fp16 windows hold exactly these values); the REFERENCE view's features are "
                                                             "additionally rounded to fp16 here (planar fp32 keeps them fp32: ~3e-6 depth difference); the gather of stages 3-4 reads "
                                                             "the taps straight from the tiles, four 16-byte buffer loads per plane and octet, no bounding box / window / barrier"}
                del sets_t, outt
                torch.cuda.empty_cache()
            except Exception as e:
                result["fp16_tiles_handoff_mode"] = {"error": repr(e)}

    # ---- extra (round 5): the SHIPPED regulariser mix (stage-1 transformer + PE3D - what released checkpoints run) on the same
    #      inputs, outside the timed headline: value + its own parity against the oracle ----
    if world == 1 and a.cost_reg != "shipped" and not a.no_shipped_leg and is_cfg2 and a.feat_layout == "planar":
        try:
            heads = build_head(device, shipped=True, conv_precision=a.conv_precision)
            n2 = max(2, a.steps // 4)
            ts, outs, issue_s = side_leg(heads, n2)
            sh = {"cost_reg_type": SHIPPED["cost_reg_type"], "value": R / ts, "unit": "ref-views/s", "ms_per_ref_view": ts / R * 1e3, "steps": n2,
                  "issue": issue_s, "attention_precision": heads.fusions[0].cost_reg.attention_precision,
                  "note": "stage 1 = PureTransformerCostReg (6 blocks, 27 648 tokens at cfg2) + Frustoconical PE, stages 2-4 as in the headline"}
            if not a.no_cpu_baseline:
                from oracle import ref_path as O
                sds = [{k: v.detach().cpu() for k, v in st.state_dict().items()} for st in heads.fusions]
                torch.set_num_threads(min(16, os.cpu_count() or 1))
                t0 = time.time()
                with torch.no_grad():
                    refs = O.cascade_forward({k: v.float().cpu() for k, v in feats.items()}, {k: v.cpu() for k, v in projs.items()}, dv.cpu(), sds,
                                             ndepths=ARGS["ndepths"], depth_interals_ratio=ARGS["depth_interals_ratio"], base_ch=ARGS["base_ch"], tmp=TMP,
                                             use_pe3d=True, transformer_config=SHIPPED["transformer_config"])
                d, r = outs["refined_depth"].cpu(), refs["refined_depth"]
                sh["parity"] = {"refined_depth_rel_l1_vs_oracle": float(((d - r).abs() / r.abs()).mean()),
                                "confidence_max_abs_vs_oracle": float((outs["photometric_confidence"].cpu() - refs["photometric_confidence"]).abs().max()),
                                "bar": 1e-3, "oracle_pass_s": time.time() - t0}
            result["shipped"] = sh
            del heads, outs
            torch.cuda.empty_cache()
        except Exception as e:
            result["shipped"] = {"error": repr(e)}

    # ---- extra (round 5): the producer-side feature emitter on cfg2's stage shapes, outside the timed region ----
    if world == 1 and not a.no_profile and is_cfg2:
        try:
            result["feature_emitter"] = emitter_leg(device, a.views)
        except Exception as e:
            result["feature_emitter"] = {"error": repr(e)}

    # ---- extra: one training step (forward + backward) per cascade stage through the native kernels ----
    if world == 1 and not a.no_train_leg and a.cost_reg != "shipped":
        try:
            result["training_step"] = training_leg(device)
        except Exception as e:  # the headline does not depend on it
            result["training_step"] = {"error": repr(e)}

    # ---- view-sharded latency mode (N > 1): source views over ranks + RCCL collectives per stage ----
    if world > 1:
        # The headline above is already measured.  The extra leg exercises collectives that have only ever run on gloo in the build
        # container: a watchdog prints the JSON line without it and ends the process if it does not come back.
        import threading
        # ... and if the process is KILLED inside the leg (an abort inside the communicator on this or another rank: torchrun then
        # terminates every worker) nothing in this interpreter gets to print: a detached helper process holds the headline line and
        # prints it to this rank's stdout if the pipe to it closes without the all-clear.
        guard = HeadlineGuard(dict(result, view_sharded={"error": "the process was terminated inside the view-sharded leg"})) if rank == 0 else None

        def _give_up():
            result["view_sharded"] = {"error": "view-sharded leg did not finish within %d s (abandoned by the watchdog)" % a.view_sharded_timeout}
            if rank == 0:
                guard.disarm()
                print(json.dumps(result), flush=True)
            os._exit(0)
        dog = threading.Timer(a.view_sharded_timeout, _give_up)
        dog.daemon = True
        dog.start()
        try:
            result["view_sharded"] = view_sharded_leg(head, a, device, world, rank, sync_all, fdt)
        except Exception as e:  # report the failure instead of dying
            result["view_sharded"] = {"error": repr(e)}
        dog.cancel()
        if guard is not None:
            guard.disarm()

    # ---- torch-ROCm composite baseline: the op-for-op restatement of the reference's
    #      PyTorch path (oracle/ref_path.py = cost_volume.py:51-133 through F.grid_sample / conv3d / conv_transpose3d / batch_norm) with its
    #      tensors on THIS GPU - what a user who runs the reference on stock PyTorch-ROCm gets - outside the timed region, N = 1 only ----
    if world == 1 and not a.no_cpu_baseline and not a.no_composite_baseline:
        try:
            result["torch_rocm_composite"] = composite_baseline(head, feats, projs, dv, out, shipped=a.cost_reg == "shipped")
        except Exception as e:
            result["torch_rocm_composite"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle on this host's cores (rank 0, N = 1 only) + a parity read-out ----
    if world == 1 and not a.no_cpu_baseline:
        planar = {k: (v.unpack() if hasattr(v, "unpack") else v) for k, v in feats.items()}
        cb, ref = cpu_baseline(head, planar, projs, dv, shipped=a.cost_reg == "shipped")
        result["cpu_baseline"] = cb
        d, r = out["refined_depth"].cpu(), ref["refined_depth"]
        result["parity"] = {"refined_depth_rel_l1_vs_oracle": float(((d - r).abs() / r.abs()).mean()),
                            "confidence_max_abs_vs_oracle": float((out["photometric_confidence"].cpu() - ref["photometric_confidence"]).abs().max()),
                            "bar": 1e-3}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        # the line is out; a rank whose peers left through their watchdog must not wait in the closing barrier for ever
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


def composite_baseline(head, feats, projs, dv, out, shipped=False):
    """The reference's own composite PyTorch path on the MI355X: oracle.ref_path.cascade_forward (op-for-op the reference's
    StageNet / CostRegNet / cascade loop, fp32, no autocast) with every tensor on cuda:0 - stock PyTorch-ROCm kernels (MIOpen
    convolutions, ATen grid_sample).  One warm-up pass (MIOpen's find step), then the median of three; the HIP path's output is
    compared with it as a second parity read-out (same device, other kernels)."""
    from oracle import ref_path as O
    dev = dv.device
    # MIOpen's default find mode spends ~2 minutes benchmarking every convolution shape of the cascade in the first pass; "FAST" (immediate mode) picks the
    # same-speed kernels here (measured: 24.9 vs 24.8 ref-views/s steady state, warm-up 0.45 s vs 124 s; scripts/gpu_r6k.sh).  Only this leg uses MIOpen
    # (the product path has no MIOpen call), and the library reads the variable at its first convolution: set here unless the caller chose a mode.
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    sds = [{k: v.detach().to(dev) for k, v in st.state_dict().items()} for st in head.fusions]
    f = {k: (v.unpack() if hasattr(v, "unpack") else v).float() for k, v in feats.items()}

    def one():
        with torch.no_grad():
            return O.cascade_forward(f, projs, dv, sds, ndepths=ARGS["ndepths"], depth_interals_ratio=ARGS["depth_interals_ratio"],
                                     base_ch=ARGS["base_ch"], tmp=TMP, use_pe3d=shipped,
                                     transformer_config=SHIPPED["transformer_config"] if shipped else None)
    t0 = time.perf_counter()
    o = one()
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = one()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[1]
    d, r = out["refined_depth"], o["refined_depth"]
    return {"value": 1.0 / med, "unit": "ref-views/s", "ms_per_ref_view": med * 1e3, "kind": "port",
            "sample": "1 reference view of the bench workload per pass (full 4-stage cascade, fp32, batch 1, one stream); warm-up pass %.2f s, "
                      "then the median of three passes: %s ms" % (warm, " / ".join("%.1f" % (t * 1e3) for t in ts)),
            "peak_memory_gb": torch.cuda.max_memory_allocated() / 1e9,
            "hip_path_vs_this_refined_depth_rel_l1": float(((d - r).abs() / r.abs()).mean()),
            "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE"),
            "note": "oracle/ref_path.py (the CPU restatement of models/cost_volume.py:51-133 + module.py regularisers + the cascade loop) with its "
                    "tensors on cuda:0: stock PyTorch-ROCm composite kernels, the baseline a patch_model user starts from (MIOpen immediate mode unless "
                    "MIOPEN_FIND_MODE is set: the exhaustive default costs a 2-minute first pass for the same steady state)"}


def emitter_leg(device, V):
    """TiledFeatureHead (mvs_conv2d3x3_tiles_fwd: the feature side's last 3x3 convolution writing bf16 octet tiles from its epilogue,
    FMT.py:195-197) on the stage 2-4 shapes of the workload, all V views per launch, and the converter pass (mvs_pack_features) it makes
    unnecessary.  Features are INPUTS of the timed path: this is not part of `value`."""
    import torch.nn as nn
    from mvsformerplusplus_amd import TiledFeatureHead, ops
    out = {"unit": "ms per reference view (all %d views of a stage)" % V, "stages": {},
           "note": "Conv2d(C, C, 3, padding=1, bias=False) fp32 planar in -> bf16 [V, C/8, H, W, 8] tiles out, split-bf16 MFMA (fp32-equivalent); "
                   "pack_features_ms = the planar -> tiled converter pass (fp32 in, bf16 out) a producer without the emitter pays on top of its convolution"}

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 * 1e3

    tot = 0.0
    for stage, C, H, W in ((2, 32, 288, 384), (3, 16, 576, 768), (4, 8, 1152, 1536)):
        conv = nn.Conv2d(C, C, 3, padding=1, bias=False)
        head = TiledFeatureHead(conv).to(device)
        x = torch.randn(1, V, C, H, W, device=device)
        ms = timed(lambda: head(x))
        nbytes = V * C * H * W * (4 + 2)
        y = torch.randn(1, V, C, H, W, device=device)
        pk = timed(lambda: ops.pack_features(y, dtype=torch.bfloat16))
        out["stages"]["stage%d" % stage] = {"C": C, "H": H, "W": W, "ms": ms, "algorithmic_gbs": nbytes / ms / 1e6,
                                            "gflop": 2.0 * 9 * C * C * V * H * W / 1e9, "pack_features_ms": pk}
        tot += ms
        del x, y
    out["ms_stages_2_to_4"] = tot
    return out


def training_leg(device):
    """Forward + backward of each cascade stage in train mode (native gather / convolution / BatchNorm kernels behind autograd
    Functions, mvsformerplusplus_amd/training.py) at DTU-training-like sizes: B = 2, V = 5, 512 x 640 at the finest stage."""
    import torch
    from mvsformerplusplus_amd import synth
    from mvsformerplusplus_amd.cost_volume import StageNet
    out = {"batch": 2, "views": 5, "unit": "ms per stage step (forward + backward)", "stages": {},
           "note": "not part of the timed headline; see DESIGN.md section 8 for the comparison with the PyTorch-autograd route"}
    B, V = 2, 5
    for stage, C, D, H, W in ((0, 64, 32, 64, 80), (1, 32, 16, 128, 160), (2, 16, 8, 256, 320), (3, 8, 4, 512, 640)):
        net = StageNet({"base_ch": [8] * 4, "depth_type": ["ce"] * 4}, D, stage).to(device).train()
        cams = synth.make_cameras(V, H, W, baseline=30.0, seed=1, batch=B).to(device)
        g = torch.Generator().manual_seed(stage)
        feats = torch.randn(B, V, C, H, W, generator=g).to(device).requires_grad_(True)
        hyp = (torch.linspace(900, 450, D)[None, :, None, None] * (1 + 0.02 * torch.rand(B, D, H, W, generator=g))).to(device).contiguous()

        def step():
            net(feats, cams, hyp, 1.0)["prob_volume_pre"].square().mean().backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        out["stages"]["stage%d" % (stage + 1)] = {"C": C, "D": D, "H": H, "W": W, "ms": (time.perf_counter() - t0) / 5 * 1e3}
        assert torch.isfinite(feats.grad).all()
    out["ms_all_stages"] = sum(v["ms"] for v in out["stages"].values())
    return out


def view_sharded_leg(head, a, device, world, rank, sync_all, fdt):
    """Latency mode of SURVEY.md section 8e on BASELINE configs[2]'s shape (V = 10: nine source views over the ranks): ONE
    reference view at a time, source views sharded over the ranks, partial cost volumes combined per stage."""
    import torch.distributed as dist
    from mvsformerplusplus_amd import synth
    V = max(a.views, 10)
    f0, p0, d0 = synth.make_cascade_inputs(a.height, a.width, V, seed=7, device=device, feat_dtype=fdt)   # same seed on every rank
    head.set_view_group(dist.group.WORLD)
    with torch.no_grad():
        for _ in range(2):
            o_sh = head(f0, p0, d0, tmp=TMP)
        sync_all()
        t0 = time.perf_counter()
        n_lat = max(3, a.steps // 4)
        for _ in range(n_lat):
            o_sh = head(f0, p0, d0, tmp=TMP)
        sync_all()
        lat = (time.perf_counter() - t0) / n_lat
        head.set_view_group(None)
        o_un = head(f0, p0, d0, tmp=TMP)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_lat):
            o_un = head(f0, p0, d0, tmp=TMP)
        torch.cuda.synchronize()
        lat_un = (time.perf_counter() - t0) / n_lat
        agree = float(((o_sh["refined_depth"] - o_un["refined_depth"]).abs() / o_un["refined_depth"].abs()).mean())
    tl = torch.tensor([lat], dtype=torch.float64, device=device)
    dist.all_reduce(tl, op=dist.ReduceOp.MAX)
    shapes = [(f0["stage%d" % (s + 1)].shape[-2:], ARGS["ndepths"][s]) for s in range(4)]
    coll = [int((8 * D + 1) * int(hw[0]) * int(hw[1]) * 4) for hw, D in shapes]
    moved = [int(st.last_collective_bytes) for st in head.fusions]
    modes = ["slab" if (st._slab_plan(int(hw[0]), world) is not None) else "allreduce" for st, (hw, D) in zip(head.fusions, shapes)]
    return {"ms_per_ref_view": float(tl.item()) * 1e3, "ref_views_per_s": 1.0 / float(tl.item()),
            "unsharded_ms_per_ref_view_one_gpu": lat_un * 1e3, "speedup_vs_one_gpu": lat_un / float(tl.item()),
            "views": V, "ranks": world, "refined_depth_rel_l1_vs_unsharded": agree,
            "partial_volume_bytes_per_stage": coll, "bytes_sent_or_reduced_per_rank_per_stage": moved, "mode_per_stage": modes,
            "note": "ONE reference view at a time, %d source views sharded over %d ranks (SURVEY.md section 8e)" % (V - 1, world)}


if __name__ == "__main__":
    main()
