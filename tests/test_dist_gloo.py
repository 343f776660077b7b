"""CPU, world_size 2, gloo: the view-sharded StageNet path (source views split over ranks, one all-reduce of
[volume_sum || vis_sum] per stage) gives every rank the single-process result.
The kernels run through the host emulator (tests/hipemu) because this container has no GPU."""
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


def _worker(rank, world, port, emu_path, V, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import golden_weights, load_golden
    from mvsformerplusplus_amd import _lib
    from mvsformerplusplus_amd.cost_volume import StageNet
    _lib._LIB = _lib.bind(emu_path)
    _lib._REQUIRE_DEVICE = False
    fx = load_golden("f2_stage_s3.npz")
    args = {"base_ch": [8] * 4, "depth_type": ["ce"] * 4}
    net = StageNet(args, 4, 3)
    net.load_state_dict(golden_weights(fx), strict=True)
    net.eval()
    feats, proj = fx["features"], fx["proj"]
    if V > feats.shape[1]:                      # more source views than ranks: repeat views with shifted cameras
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
    gathered = [torch.zeros_like(sharded["depth"]) for _ in range(world)]
    dist.all_gather(gathered, sharded["depth"])
    same = all(torch.equal(gathered[0], g) for g in gathered)
    q.put((rank, err, same))
    dist.destroy_process_group()


@pytest.mark.parametrize("V", [3, 4, 2])
def test_view_sharded_stage_matches_single_process(V):
    import hipemu_build
    emu = hipemu_build.build()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu, V, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same in res:
        assert err <= 1e-5, "rank %d: sharded depth differs from single-process depth by %g" % (rank, err)
        assert same, "ranks disagree after the all-reduce"
