import sys, time, tempfile, shutil, torch
sys.path.insert(0, '/root/repo')
from cartpolesimulation_amd.configs import legacy_mppi_config
from cartpolesimulation_amd.engine import MPPIEngine
from cartpolesimulation_amd import recording as R
from cartpolesimulation_amd import harness as Hn
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = MPPIEngine(E, legacy_mppi_config(num_rollouts=3500, mpc_horizon=35))
out = tempfile.mkdtemp()
# time the device loop alone
import numpy as np
rng = np.random.Generator(np.random.SFC64(0))
s0 = Hn.generate_random_initial_states(E, rng, eng.phys.TrackHalfLength)
exp = Hn.BatchedCartPoleExperiment(eng, seed=0)
exp.run(s0, 20); torch.cuda.synchronize()
t0 = time.perf_counter(); res = exp.run(s0, 500); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"device loop: {E} envs x 500 control steps (10 s each) in {t1 - t0:.2f} s = {E * 500 / (t1 - t0):.3g} control steps/s")
import os
for native in (True, False):
    t0 = time.perf_counter(); paths = R.generate_dataset(eng, E, 10.0, out, seed=0, native=native); t1 = time.perf_counter()
    print(f"generate_dataset total ({'native writer' if native else 'python csv module'}): {t1 - t0:.2f} s for {len(paths)} files "
          f"({sum(os.path.getsize(p) for p in paths) / 1e6:.1f} MB)")
# the writer alone, on the recording of the run above, by thread count
Lv = np.full(E, eng.phys.L, np.float32)
block = R._host_block(res, Lv, eng.phys)
header = R.create_csv_header(10.0, 0.002, 0.02, 0.02, "mpc", "mppi", eng.phys)
for nt in (1, 4, 16, 0):
    paths = [os.path.join(out, f"w{nt}_{e}.csv") for e in range(E)]
    t0 = time.perf_counter()
    R.write_recordings_native(paths, block, 0.02, np.zeros(E, np.float32), np.ones(E, np.float32), Lv, eng.phys, header, n_threads=nt)
    print(f"cpmppi_write_recordings alone, n_threads={nt or 'auto'}: {time.perf_counter() - t0:.3f} s")
shutil.rmtree(out)
