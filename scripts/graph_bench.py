"""Experiment: is the 3-stream bench host-bound?  (1) host time to ISSUE one reference view (no sync), (2) throughput when each
(stream, input set) pair replays a captured hipGraph instead of being issued from Python."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import synth
dev = torch.device("cuda:0")
head = bench.build_head(dev)
nsets = 4
sets = [synth.make_cascade_inputs(1152, 1536, 5, seed=i, device=dev) for i in range(nsets)]
with torch.no_grad():
    for i in range(6):
        head(*sets[i % nsets], tmp=bench.TMP)
    torch.cuda.synchronize()
    # (1) host issue time: queue 16 views on one stream, clock the Python loop only
    t0 = time.perf_counter()
    for i in range(16):
        head(*sets[i % nsets], tmp=bench.TMP)
    t_issue = (time.perf_counter() - t0) / 16
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 16
    print("host issue time %.3f ms / ref view; with completion %.3f ms / ref view (one stream)" % (t_issue * 1e3, t_total * 1e3))
    for nstreams in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        graphs = []
        for s, st in enumerate(streams):
            row = []
            for k in range(nsets):
                g = torch.cuda.CUDAGraph()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    head(*sets[k], tmp=bench.TMP)
                torch.cuda.current_stream().wait_stream(st)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=st):
                    o = head(*sets[k], tmp=bench.TMP)
                row.append((g, o))
            graphs.append(row)
        torch.cuda.synchronize()
        def run(n):
            for j in range(n):
                st = streams[j % nstreams]
                with torch.cuda.stream(st):
                    graphs[j % nstreams][j % nsets][0].replay()
        run(32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(320)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 320
        print("graph replay, %d stream(s): %.3f ms / ref view = %.1f ref-views/s" % (nstreams, dt * 1e3, 1 / dt))
        del graphs
