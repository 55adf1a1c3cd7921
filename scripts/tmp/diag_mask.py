import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import rel_l2
from oracle.pipeline_ref import synthetic_inputs
from oracle.pixart_ref import PixArtConfig, PixArtOracle, make_weights
from test_gpu_plugins import _model
layers, steps, B = 2, 1, 2
bits = np.ones((steps, layers, 3), dtype=bool)
inp = synthetic_inputs(B)
enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0)
mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0)
g = torch.Generator().manual_seed(21)
self_mask = (torch.rand(2 * B, 256, generator=g) > 0.3).to(torch.int64); self_mask[:, 0] = 1
soft = ((1 - mask.float()) * -10000.0 + torch.randn(2 * B, 120, generator=g) * 0.5)[:, None, :].half()
hard3 = ((1 - mask.float()) * -10000.0)[:, None, :].half()
x = torch.randn(2 * B, 4, 32, 32, generator=g); t = torch.full((2 * B,), 900.0)
for name, am, em in [("plain 2-D enc mask", None, mask), ("hard 3-D enc bias", None, hard3), ("soft 3-D enc bias", None, soft), ("self mask only", self_mask, mask), ("self 3-D bias", ((1 - self_mask.float()) * -10000.0)[:, None, :].half(), mask), ("both", self_mask, soft)]:
    tr, sched, cfg, w = _model(layers, steps, bits, {})
    ref = PixArtOracle(cfg, w)
    out = tr(x.cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=em.cuda(), attention_mask=None if am is None else am.cuda(), timestep=t.cuda(), return_dict=False)[0]
    exp = ref.forward(x.half().float(), enc.half().float(), em.float() if em.ndim == 3 else em, t, 0, bits[0], attention_mask=None if am is None else (am.float() if am.ndim == 3 else am))
    print(f"{name:22s} rel-L2 {rel_l2(out, exp):.3e}", flush=True)
    tr.engine.close()
