import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
ctx = {"world": 1, "rank": 0, "local_rank": 0, "device": torch.device("cuda", 0), "collective": False, "backend": "nccl"}
torch.cuda.set_device(0)
for name, (E, N, H) in (("C4", (64, 2048, 50)), ("C3", (64, 4096, 100)), ("E1", (1, 1024, 50))):
    w = bench.Workload(ctx, E, N, H)
    for prof in (1, 8, 0, 1, 8, 0):
        for i in range(10): w.step(i)
        w.barrier()
        w.eng.set_profiling(bool(prof), group=prof if prof else 1)
        t0 = time.perf_counter()
        K = 200
        for i in range(K): w.step(10 + i)
        w.barrier()
        dt = time.perf_counter() - t0
        k_ms = None
        if prof:
            r, _ = w.eng.get_profile(); k_ms = float(np.mean(r))
        w.eng.set_profiling(False)
        print(name, f"event group {prof}" if prof else "no events", f"{dt/K*1e6:.1f} us/step", f"kernel {k_ms*1e3:.1f} us" if k_ms else "", flush=True)
    w.close()
