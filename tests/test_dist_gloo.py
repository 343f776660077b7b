"""CPU, gloo, world_size 2 and 4: the view-sharded latency mode gives every rank the single-process
result - one StageNet in all-reduce mode, one StageNet in slab mode with a real interior halo cut, the whole 4-stage cascade,
and an uneven 9-source-views-over-4-ranks split.  The kernels run through the host emulator (tests/hipemu) because this
container has no GPU; the same Python code drives RCCL on the GPU node (backend "nccl")."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, emu_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest  # noqa: F401
    from mvsformerplusplus_amd import _lib
    _lib._LIB = _lib.bind(emu_path)
    _lib._REQUIRE_DEVICE = False


def _agree(t, world):
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t.contiguous())
    return all(torch.equal(gathered[0], g) for g in gathered)


def _stage_worker(rank, world, port, emu_path, V, precision, q):
    _setup(rank, world, port, emu_path)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd.cost_volume import StageNet
    fx = load_golden("f2_stage_s3.npz")
    args = dict({"base_ch": [8] * 4, "depth_type": ["ce"] * 4}, **({"conv_precision": precision} if precision else {}))
    net = StageNet(args, 4, 3)
    net.load_state_dict(golden_weights(fx), strict=True)
    net.eval()
    feats, proj = fx["features"], fx["proj"]
    if V > feats.shape[1]:                      # more source views than the fixture holds: repeat views with shifted cameras
        reps = (V + feats.shape[1] - 1) // feats.shape[1]
        feats = feats.repeat(1, reps, 1, 1, 1)[:, :V].contiguous()
        proj = proj.repeat(1, reps, 1, 1, 1)[:, :V].clone()
        for v in range(V):
            proj[:, v, 0, 0, 3] += 3.0 * v
    with torch.no_grad():
        single = net(feats, proj, fx["hyp"], 1.0)
        net.view_group = dist.group.WORLD
        sharded = net(feats, proj, fx["hyp"], 1.0)
    err = float((single["depth"] - sharded["depth"]).abs().max() / single["depth"].abs().max())
    q.put((rank, err, _agree(sharded["depth"], world)))
    dist.destroy_process_group()


def _slab_worker(rank, world, port, emu_path, precision, q):
    """One fine-stage StageNet (C = 8, D = 4, CostRegNet3D) on a 208 x 16 map: slabs of 104 rows, halo-extended to [0,144) and
    [64,208) - the regulariser really runs on cut volumes whose first / last 40 rows are discarded."""
    _setup(rank, world, port, emu_path)
    from mvsformerplusplus_amd import synth
    from mvsformerplusplus_amd.cost_volume import StageNet
    H, W, V, D = 208, 16, 3, 4
    net = StageNet(dict({"base_ch": [8] * 4, "depth_type": ["ce"] * 4}, **({"conv_precision": precision} if precision else {})), D, 3)
    net.load_state_dict(synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), 3), strict=True)
    net.eval()
    g = torch.Generator().manual_seed(5)
    cams = synth.make_cameras(V, H, W, baseline=30.0, seed=2)
    feats = torch.randn(1, V, 8, H, W, generator=g)
    hyp = (torch.linspace(800, 500, D)[None, :, None, None] * (1 + 0.02 * torch.rand(1, D, H, W, generator=g))).contiguous()
    with torch.no_grad():
        single = net(feats, cams, hyp, 1.0)
        net.view_group = dist.group.WORLD
        net.shard_mode = "auto"
        assert net._slab_plan(H, world) is not None, "208 rows over 2 ranks must select the slab form"
        sharded = net(feats, cams, hyp, 1.0)
    errs = [float((single[k] - sharded[k]).abs().max() / single[k].abs().max()) for k in ("depth", "photometric_confidence", "prob_volume", "prob_volume_pre")]
    q.put((rank, max(errs), _agree(sharded["depth"], world) and _agree(sharded["prob_volume"], world)))
    dist.destroy_process_group()


def _other_groups_worker(rank, world, port, emu_path, mode, q):
    """A stage with base_ch = 4 (fixture F15: the shape-generic regulariser, fp32 partial volumes of 4 groups) sharded over the ranks:
    all-reduce form and the slab exchange forced on a 32-row map (the halo covers everything: every rank regularises the whole map)."""
    _setup(rank, world, port, emu_path)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd.cost_volume import StageNet
    fx = load_golden("f15_stage_g4_s3.npz")
    net = StageNet({"base_ch": [4] * 4, "depth_type": ["ce"] * 4, "conv_precision": "bf16x3"}, fx["hyp"].shape[1], 3)
    net.load_state_dict(golden_weights(fx), strict=True)
    net.eval()
    with torch.no_grad():
        single = net(fx["features"], fx["proj"], fx["hyp"], 1.0)
        net.view_group = dist.group.WORLD
        net.shard_mode = mode
        sharded = net(fx["features"], fx["proj"], fx["hyp"], 1.0)
    errs = [float((single[k] - sharded[k]).abs().max() / single[k].abs().max()) for k in ("depth", "photometric_confidence", "prob_volume_pre")]
    ref = float((sharded["depth"] - fx["depth"]).abs().div(fx["depth"].abs()).mean())
    q.put((rank, max(max(errs), ref), _agree(sharded["depth"], world)))
    dist.destroy_process_group()


def _cascade_worker(rank, world, port, emu_path, V, mode, q):
    _setup(rank, world, port, emu_path)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd.cascade import CascadeDepthHead
    fx = load_golden("f4_cascade.npz")
    args = {"base_ch": [8] * 4, "depth_type": ["ce"] * 4, "fusion_type": "cnn", "cost_reg_type": ["Normal"] * 4,
            "ndepths": [32, 16, 8, 4], "depth_interals_ratio": [4.0, 2.67, 1.5, 1.0], "inverse_depth": True}
    head = CascadeDepthHead(args)
    for s in range(4):
        head.fusions[s].load_state_dict(golden_weights(fx, "w%d." % (s + 1)), strict=True)
    head.eval()
    feats = {"stage%d" % s: fx["features%d" % s] for s in range(1, 5)}
    projs = {"stage%d" % s: fx["proj%d" % s] for s in range(1, 5)}
    if V > feats["stage1"].shape[1]:
        for k in feats:
            n = feats[k].shape[1]
            reps = (V + n - 1) // n
            feats[k] = feats[k].repeat(1, reps, 1, 1, 1)[:, :V].contiguous()
            projs[k] = projs[k].repeat(1, reps, 1, 1, 1)[:, :V].clone()
            for v in range(V):
                projs[k][:, v, 0, 0, 3] += 2.0 * v
    with torch.no_grad():
        single = head(feats, projs, fx["depth_values"])
        head.set_view_group(dist.group.WORLD, shard_mode=mode)
        sharded = head(feats, projs, fx["depth_values"])
    err = float(((single["refined_depth"] - sharded["refined_depth"]).abs() / single["refined_depth"].abs()).mean())
    ok = _agree(sharded["refined_depth"], world) and _agree(sharded["photometric_confidence"], world)
    q.put((rank, err, ok))
    dist.destroy_process_group()


def _run(target, world, *args):
    import hipemu_build
    emu = hipemu_build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, emu) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("precision,tol", [(None, 5e-4), ("bf16x3", 1e-5)])
@pytest.mark.parametrize("V", [4, 2])        # 3 source views over 2 ranks (2 + 1) and 1 source view (one rank idle); the even 1 + 1 split is inside the cascade tests
def test_view_sharded_stage_matches_single_process(V, precision, tol):
    """precision None = the product default (policy "stagemix": this D = 4 stage runs "f16mix"): the all-reduced fp32 volume is rounded to fp16 once (mvs_volume_to_f16) where the
    single-process aggregate pass rounds its own sum - the same values up to fp32 summation order, i.e. an fp16 ulp at a few voxels."""
    for rank, err, same in _run(_stage_worker, 2, V, precision):
        assert err <= tol, "rank %d: sharded depth differs from single-process depth by %g" % (rank, err)
        assert same, "ranks disagree after the all-reduce"


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-5), (None, 2e-3)])
def test_slab_mode_matches_single_process(precision, tol):
    """None = the product default (an fp16-format stage here): the sharded path rounds the SUMMED partial volume to fp16 once (mvs_volume_to_f16), the single-process
    path rounds in the aggregate pass - the same values up to fp32 summation order, i.e. an fp16 ulp (5e-4) at a few voxels."""
    for rank, err, same in _run(_slab_worker, 2, precision):
        assert err <= tol, "rank %d: slab-sharded outputs differ from single-process outputs by %g" % (rank, err)
        assert same, "ranks disagree after the slab all-gather"


@pytest.mark.parametrize("mode", ["slab"])        # the all-reduce form is G-agnostic (one flat buffer); the slab exchange kernels carry G
def test_view_sharded_stage_other_base_ch(mode):
    for rank, err, same in _run(_other_groups_worker, 2, mode):
        assert err <= 2e-5, "rank %d: sharded base_ch = 4 stage differs from the single-process / reference result by %g" % (rank, err)
        assert same, "ranks disagree"


@pytest.mark.parametrize("world,V,mode", [(4, 10, "auto")])
def test_view_sharded_cascade_matches_single_process(world, V, mode):
    """The whole 4-stage cascade on the uneven BASELINE configs[2] split - 9 source views over 4 ranks (3 + 2 + 2 + 2) in "auto" mode.  (2 ranks
    with the slab exchange forced on every stage, and 9 views over 8 ranks, run inside the two-views test below, which also compares the
    sharded result with the single-process one.)"""
    for rank, err, same in _run(_cascade_worker, world, V, mode):
        assert err <= 1e-4, "rank %d: sharded refined depth rel-L1 %g vs the single-process cascade" % (rank, err)
        assert same, "ranks hold different results"


def _two_views_worker(rank, world, port, emu_path, V, mode, q):
    """Two reference views through ONE view-sharded head back to back (the schedule of two views in flight per group: every rank
    issues A, B, A): the persistent scratch of the exchange is reused across views, so A must come out the same both times,
    B must not be disturbed by A's buffers, and the ranks must agree on all three.  V > the fixture's view count: views repeated with
    shifted cameras (BASELINE configs[2]: 9 source views over 8 ranks = one rank carries two)."""
    _setup(rank, world, port, emu_path)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd.cascade import CascadeDepthHead
    fx = load_golden("f4_cascade.npz")
    args = {"base_ch": [8] * 4, "depth_type": ["ce"] * 4, "ndepths": [32, 16, 8, 4], "depth_interals_ratio": [4.0, 2.67, 1.5, 1.0], "inverse_depth": True}
    head = CascadeDepthHead(args)
    for s in range(4):
        head.fusions[s].load_state_dict(golden_weights(fx, "w%d." % (s + 1)), strict=True)
    head.eval()
    fa = {"stage%d" % s: fx["features%d" % s] for s in range(1, 5)}
    projs = {"stage%d" % s: fx["proj%d" % s] for s in range(1, 5)}
    if V > fa["stage1"].shape[1]:
        for k in fa:
            n = fa[k].shape[1]
            reps = (V + n - 1) // n
            fa[k] = fa[k].repeat(1, reps, 1, 1, 1)[:, :V].contiguous()
            projs[k] = projs[k].repeat(1, reps, 1, 1, 1)[:, :V].clone()
            for v in range(V):
                projs[k][:, v, 0, 0, 3] += 2.0 * v
    g = torch.Generator().manual_seed(4)
    fb = {k: v + 0.3 * torch.randn(v.shape, generator=g) for k, v in fa.items()}        # a second reference view: other features, same rig
    with torch.no_grad():
        single_b = head(fb, projs, fx["depth_values"])["refined_depth"]
        head.set_view_group(dist.group.WORLD, shard_mode=mode)
        a1 = head(fa, projs, fx["depth_values"])["refined_depth"].clone()
        b1 = head(fb, projs, fx["depth_values"])["refined_depth"].clone()
        a2 = head(fa, projs, fx["depth_values"])["refined_depth"].clone()
    err = float(((single_b - b1).abs() / single_b.abs()).mean())
    ok = torch.equal(a1, a2) and _agree(a1, world) and _agree(b1, world) and _agree(a2, world)
    q.put((rank, err, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,V,mode", [(2, 4, "slab"), (8, 10, "auto")])
def test_two_views_in_flight_schedule_is_repeatable_and_rank_identical(world, V, mode):
    """(2 ranks, slab exchange forced) and BASELINE configs[2]'s split: 9 source views over 8 ranks (one rank carries two views, seven carry
    one; product-default precision), two reference views in flight through one sharded head."""
    for rank, err, ok in _run(_two_views_worker, world, V, mode):
        assert err <= 3e-4, "rank %d: second view's sharded depth differs from its single-process depth by %g" % (rank, err)
        assert ok, "views issued back to back through one sharded head disturbed each other or the ranks disagree"


def _syncbn_worker(rank, world, port, emu_path, q):
    """Training path under SyncBatchNorm: each rank holds one sample of a batch of two; the BatchNorm statistics (forward and
    backward sums) are all-reduced inside the native kernels' host code, so every rank's feature gradient equals the matching
    sample's gradient of a single-process batch-of-two step, and the parameter gradients summed over the ranks equal its."""
    _setup(rank, world, port, emu_path)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd.cost_volume import StageNet
    fx = load_golden("f12_train_backward_s3.npz")
    args = {"base_ch": [8] * 4, "depth_type": ["ce"] * 4}

    def make():
        net = StageNet(args, 4, 3)
        net.load_state_dict(golden_weights(fx), strict=True)
        return net.train()

    def loss_of(out, R):
        return (out["prob_volume"] * R).sum() + 0.05 * out["prob_volume_pre"].pow(2).sum()

    full = make()
    f_all = fx["features"].clone().requires_grad_(True)
    loss_of(full(f_all, fx["proj"], fx["hyp"], 1.0), fx["R"]).backward()
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(make())
    sl = slice(rank, rank + 1)
    f_loc = fx["features"][sl].clone().requires_grad_(True)
    loss_of(net(f_loc, fx["proj"][sl], fx["hyp"][sl], 1.0), fx["R"][sl]).backward()
    err = float((f_loc.grad - f_all.grad[sl]).abs().max() / f_all.grad.abs().max())
    if os.environ.get("MVS_TEST_VERBOSE") and rank == 0:
        print("features", err)
    worst = 0.0
    for (n, p), (_, pf) in zip(net.named_parameters(), full.named_parameters()):
        g = p.grad.clone()
        dist.all_reduce(g)
        e = float((g - pf.grad).abs().max() / pf.grad.abs().max().clamp_min(1e-12))
        if os.environ.get("MVS_TEST_VERBOSE") and rank == 0:
            print("%-36s %.2e" % (n, e))
        worst = max(worst, e)
    stats = max(float((b - bf).abs().max()) for (n, b), (_, bf) in zip(net.named_buffers(), full.named_buffers()) if "running_" in n)
    q.put((rank, max(err, worst), stats <= 1e-4))
    dist.destroy_process_group()


def test_training_path_syncbn_matches_single_process():
    for rank, err, stats_ok in _run(_syncbn_worker, 2):
        assert err <= 2e-3, "rank %d: SyncBatchNorm training gradients differ from the single-process batch by %g" % (rank, err)
        assert stats_ok, "running statistics differ"
