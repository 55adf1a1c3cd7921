import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import rel_l2
from oracle.pipeline_ref import synthetic_inputs
from oracle.pixart_ref import PixArtConfig, PixArtOracle, make_weights
from test_gpu_plugins import _model
layers, steps, B = 2, 3, 2
bits = np.ones((steps, layers, 3), dtype=bool)
bits[1, 0, 0] = bits[1, 1, 1] = bits[2, 0, 2] = False
inp = synthetic_inputs(B)
enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0)
mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0)
g = torch.Generator().manual_seed(21)
self_mask = (torch.rand(2 * B, 256, generator=g) > 0.3).to(torch.int64); self_mask[:, 0] = 1
soft = ((1 - mask.float()) * -10000.0 + torch.randn(2 * B, 120, generator=g) * 0.5)[:, None, :].half()
xs = [torch.randn(2 * B, 4, 32, 32, generator=g) for _ in range(steps)]
for name, am, em in [("no masks", None, mask), ("self mask only", self_mask, mask), ("soft enc only", None, soft), ("both", self_mask, soft)]:
    tr, sched, cfg, w = _model(layers, steps, bits, {})
    ref = PixArtOracle(cfg, w)
    errs = []
    for s in range(steps):
        t = torch.full((2 * B,), 900.0 - 70 * s)
        out = tr(xs[s].cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=em.cuda(), attention_mask=None if am is None else am.cuda(), timestep=t.cuda(), return_dict=False)[0]
        exp = ref.forward(xs[s].half().float(), enc.half().float(), em.float() if em.ndim == 3 else em, t, s, bits[s], attention_mask=am)
        errs.append(rel_l2(out, exp))
        sched.per_step_callback(s, 0)
    print(f"{name:16s}", " ".join(f"{e:.3e}" for e in errs), tr.engine.stats()["dynamic_step_plans"], flush=True)
    tr.engine.close()
