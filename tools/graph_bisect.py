import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "maskcyclegan-vc_amd"))
    import torch
    from bench import build_nets, synthetic_batches
    from mask_cyclegan_vc.engine import TrainEngine
    conc, aux = sys.argv[1] == "1", sys.argv[2] == "1"
    dev = torch.device("cuda", 0)
    eng = TrainEngine(build_nets(dev), 1, 64)
    eng.concurrent, eng.aux_wgrad, eng.use_graphs = conc, aux, True
    bt = synthetic_batches(4, 1, 64, 0, dev)
    for i in range(6): eng.step(*bt[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): eng.step(*bt[i % 4]); eng.losses()
    torch.cuda.synchronize()
    print("conc=%s aux=%s graphs=%s: %.2f ms/step g_loss=%.4f" % (conc, aux, eng.use_graphs, 1e3 * (time.perf_counter() - t0) / 20, eng.losses()["g_loss"]))
else:
    for c, a, only in (("1", "1", "0"), ("1", "1", "1"), ("1", "1", "0,1")):
        env = dict(os.environ, MCVC_AUX_LANES=only)
        r = subprocess.run([sys.executable, __file__, c, a], capture_output=True, text=True, timeout=200, env=env)
        print("lanes=%s aux=%s aux_lanes=%s rc=%d :: %s" % (c, a, only, r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1]))
        if r.returncode:
            print(r.stderr[-600:])
