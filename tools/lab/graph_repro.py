"""tools/lab/graph_repro.py [A | A_thread | fixedinput | prewarm3 | A_sync | A_dummy | A_side]: why the NETWORK part of the
forward is not run as a hipGraph (torch.cuda.CUDAGraph) although it would shorten the end-to-end step by ~8 %
(tools/lab/graph_fwd.py: 3.61 -> 3.32 ms at B=4).

Mode A: a graph of the network at B=2 is captured on the third B=2 call; batches of B=3 run eagerly in between.  From
the second round on every replay of the B=2 graph returns garbage (1e13 .. 1e32) although its input buffers are right
and an EAGER pass on the same buffers is right; a second replay after an eager pass at the graph's own shapes is right
again; host synchronisation, an extra stream or events change nothing.  With the channels-last path off
(PPScatter.channels_last_inference = False: NCHW convolutions) it does not happen.  Reading: the NHWC implicit-GEMM
convolutions of this MIOpen are launched from mutable invoker state that a captured kernel node references instead of
copying; an eager convolution afterwards rewrites it.  A_thread (capture on a parked thread of its own = its own
MIOpen handle) survives mode A -- but eager passes at the graph's OWN shapes on other buffers (another pipeline in the
process) still change what a replay computes (O(0.3) differences instead of 1e-7).  No way to fence that from inside
this package: forward_pipelined keeps launching the network eagerly."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pp_amd  # noqa
from pp_amd import synth
from pp_amd.pipeline import PillarPipeline
from pp_amd.voxelizer import VoxelConfig, PillarVoxelizer
gpu = torch.device("cuda", 0)
MODE = sys.argv[1] if len(sys.argv) > 1 else "A"
pipe = PillarPipeline(VoxelConfig.square(16.0, 0.2, 4000, 32), feature_channels=64, device=gpu, seed=0)
pipe.model.eval()
m = pipe.model
vox = pipe.voxelizer
cfg = pipe.vox_cfg
shapes = [(15000, 3, 2), (9000, 40, 2), (12000, 80, 2), (15000, 5, 2), (7000, 9, 2), (15000, 3, 3), (11000, 60, 3),
          (9000, 61, 3), (8000, 62, 3), (15000, 7, 2), (9000, 8, 2)]
clouds = [torch.from_numpy(np.stack([synth.lidar_like(n, 16.0, sd + s) for s in range(B)])).to(gpu) for n, sd, B in shapes]
bufs = {B: (torch.empty((B, 9, cfg.max_pillars, cfg.max_points_per_pillar), dtype=torch.float32, device=gpu),
            torch.empty((B, cfg.max_pillars, 3), dtype=torch.int64, device=gpu)) for B in (2, 3)}
if MODE == "prewarm3":          # the B=3 shapes are known to MIOpen / the allocator BEFORE the capture
    with torch.no_grad():
        vox(clouds[5], out=bufs[3]); m(*bufs[3]); torch.cuda.synchronize()
g2 = out2 = None
calls2 = 0
with torch.no_grad():
    for rnd in range(5):
        res = []
        for k, c in enumerate(clouds):
            B = c.shape[0]
            if MODE == "fixedinput" and g2 is not None and B == 2:
                pass                                  # the graph's input stays what it was at capture
            else:
                vox(c, out=bufs[B])
            if B == 3:
                o = m(*bufs[3])
                res.append("%.3g" % float(o[0].abs().max()))
                del o
                continue
            calls2 += 1
            if g2 is None and calls2 <= 2:
                o = m(*bufs[2])
            else:
                if g2 is None and MODE == "A_thread":
                    # capture on ANOTHER thread (kept alive): PyTorch hands every thread its own MIOpen handle
                    import threading
                    box = {}
                    park = threading.Event()

                    def cap():
                        torch.cuda.set_device(gpu)
                        with torch.no_grad():
                            m(*bufs[2]); torch.cuda.synchronize()        # this handle's own warm-up
                            gg = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(gg):
                                oo = m(*bufs[2])
                        box["g"], box["o"] = gg, oo
                        box["done"].set()
                        park.wait()                                       # never returns the handle to the pool

                    box["done"] = threading.Event()
                    th = threading.Thread(target=cap, daemon=True)
                    th.start()
                    box["done"].wait()
                    g2, out2 = box["g"], box["o"]
                if g2 is None:
                    g2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g2):
                        out2 = m(*bufs[2])
                if MODE == "A_sync":
                    torch.cuda.synchronize()
                if MODE == "A_dummy":
                    bufs[2][1].add_(0)
                if MODE == "A_side":
                    main = torch.cuda.current_stream(); side = globals().setdefault("_side", torch.cuda.Stream())
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        g2.replay()
                    main.wait_stream(side)
                elif MODE == "A_nonnull":
                    pass
                else:
                    g2.replay()
                o = out2
            cl = o[0].clone()
            torch.cuda.synchronize()
            res.append("%.3g" % float(cl.abs().max()))
            del cl
        print(MODE, "round", rnd, res)

    # --- now the graph is bad (mode A): what makes it right again?
    if MODE == "A_never":
        def rp(tag):
            g2.replay(); torch.cuda.synchronize()
            e = m(*bufs[2]); torch.cuda.synchronize()
            print("   %-70s replay max %.3g   eager max %.3g" % (tag, float(out2[0].abs().max()), float(e[0].abs().max())))
        rp("replay again, input untouched")
        saved = (bufs[2][0].clone(), bufs[2][1].clone())
        bufs[2][0].zero_(); bufs[2][1].zero_(); torch.cuda.synchronize()
        rp("input zeroed by torch")
        bufs[2][0].copy_(saved[0]); bufs[2][1].copy_(saved[1]); torch.cuda.synchronize()
        rp("input restored by torch copy_")
        vox(clouds[0], out=bufs[2]); torch.cuda.synchronize()
        rp("input rewritten by the voxelizer (cloud 0)")
        v2 = PillarVoxelizer(cfg, device=gpu)
        v2(clouds[0], out=bufs[2]); torch.cuda.synchronize()
        rp("input rewritten by a FRESH voxelizer (cloud 0)")
        p, i = v2(clouds[1]); bufs[2][0].copy_(p); bufs[2][1].copy_(i); torch.cuda.synchronize()
        rp("input = fresh voxelizer's own output, copied in by torch (cloud 1)")
