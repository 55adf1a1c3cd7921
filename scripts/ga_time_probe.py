"""What a new schedule costs the GA evaluator on one GPU (python scripts/ga_time_probe.py [fold]): set_schedule, the first run_schedule of a batch of 16 (plan lowering +
hipGraph capture + instantiate + the run) against the steady run, graph and eager, on seed candidates of tests/golden/ga_seed_population.npz.  Round 6: first - steady =
7-27 ms without the LayerNorm fold (+ 180-280 ms with it: the folded weights of a new schedule); the 1.05 s per candidate of BASELINE config 4 is the compute of seven
batches of these lightly cached schedules (145-170 ms each), not plan building."""
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import bench
bench.torch = torch
from ecad_amd.engine import EcadEngine
from ecad_amd.pipelines.dpm_solver import DPMSolverMultistepScheduler
device = torch.device('cuda', 0)
eng = EcadEngine(num_layers=28, max_batch=32, max_tokens=256, max_text_tokens=120, dtype=torch.float16, device=0)
weights = bench.device_weights(type("C", (), dict(num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=8, patch_size=2, num_layers=28, caption_channels=4096)), seed=0, device=device)
eng.load_state_dict(weights)
solver = DPMSolverMultistepScheduler(); solver.set_timesteps(20)
ts, coefs = solver.timesteps.numpy().astype(np.float32), solver.coefficient_table()
e, m, l = bench.synthetic_batch(16, seed=1, device=device)
fold = len(sys.argv) > 1 and sys.argv[1] == 'fold'
eng.set_ln_fold(fold)
print('ln_fold', fold)
seeds = np.unpackbits(np.load('tests/golden/ga_seed_population.npz')['bits'], axis=1)[:, :1680].astype(bool)
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return 1000 * (time.perf_counter() - t0)
for c in range(3):
    bits = seeds[c].reshape(20, 28, 3)
    a = t(lambda: eng.set_schedule(bits))
    b = t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=True))
    b2 = t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=True))
    b2 = min(b2, t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=True)), t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=True)))
    print(f"graph : set_schedule {a:.1f} ms, first run {b:.1f} ms, second run {b2:.1f} ms")
for c in range(3, 6):
    bits = seeds[c].reshape(20, 28, 3)
    a = t(lambda: eng.set_schedule(bits))
    b = t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=False))
    b2 = t(lambda: eng.run_schedule(e, m, l, ts, coefs, 4.5, use_graph=False))
    print(f"eager : set_schedule {a:.1f} ms, first run {b:.1f} ms, second run {b2:.1f} ms")
print(eng.stats()["plan_builds"], eng.stats()["graph_captures"])
