"""-m gpu: the reference's per-block compute plug-in contract on the native engine, the mid-run ``reset_cache`` semantics
(row a9) and the plan-set cache, all through the C ABI.

Plug-ins: a schedule row may name ``custom_compute_attn`` / ``custom_compute_ff`` per block (``pixart_cache_schedule.py:29-37``); the block looks the name up
in ``ComputeAttnRegistry`` / ``ComputeFFRegistry`` and calls ``func(block, attn, hidden, enc, mask, **kwargs)`` / ``func(block, norm_hidden, **kwargs)``
(``cached_transformer_block.py:125-165``).  Here such a block runs as native sub-ops (LN + modulate, gate, residual, the block's own attn / ff kernels)
around the registered Python callable.  Tolerances as in test_gpu_engine.py (fp16 step vs fp32 oracle: 2e-3).
"""
import numpy as np
import pytest
import torch

from gpu_util import rel_l2, within

pytestmark = pytest.mark.gpu

from oracle.pipeline_ref import synthetic_inputs  # noqa: E402  (checker only)
from oracle.pixart_ref import PixArtConfig, PixArtOracle, make_weights  # noqa: E402


def _model(layers, steps, bits, custom, unit_gates=False):
    """Facade + schedule whose rows name plug-ins: custom = {(step, block): {"custom_compute_ff": {...}, ...}}.  unit_gates: scale_shift_table ~ N(0, 1) as in
    test_gpu_engine.unit_gate_weights (gates of O(1): a sub-layer's output reaches the prediction at full size)."""
    from ecad_amd.schedulers.cache_scheduler.pixart_cache_schedule import PixArtCacheSchedule
    from ecad_amd.schedulers.dit_scheduler.dit_scheduler import DiTScheduler, PixArtDiTSchedule
    from ecad_amd.transformer_2d_models.pixart_transformer_2d_edited import PixArtTransformer2DEdited

    sched = PixArtCacheSchedule.from_numpy(bits, "plug", num_blocks=layers, num_inference_steps=steps)
    for (s, b), extra in custom.items():
        sched.schedule[s][str(b)].update(extra)
    cfg = PixArtConfig(num_layers=layers)
    w = make_weights(cfg, seed=3)
    if unit_gates:
        g = torch.Generator().manual_seed(103)
        for k in w:
            if k.endswith("scale_shift_table"):
                w[k] = torch.randn(w[k].shape, generator=g)
    tr = PixArtTransformer2DEdited(num_layers=layers, sample_size=32, max_batch=4)
    tr.load_state_dict(w)
    tr._post_init(DiTScheduler(PixArtDiTSchedule(layers, steps, "default")), sched)
    return tr, sched, cfg, w


def _run(tr, sched, ref, bits, steps, B=2, reset_before=None):
    inp = synthetic_inputs(B)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0)
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0)
    g = torch.Generator().manual_seed(7)
    worst = 0.0
    enc_d, mask_d = enc.cuda(), mask.cuda()
    for s in range(steps):
        if reset_before is not None and s == reset_before:
            tr.reset_cache()
            ref.reset_cache()
        x = torch.randn(2 * B, 4, 32, 32, generator=g)
        t = torch.full((2 * B,), 950.0 - 60 * s)
        out = tr(x.cuda(), encoder_hidden_states=enc_d, encoder_attention_mask=mask_d, timestep=t.cuda(), return_dict=False)[0]
        exp = ref.forward(x.half().float(), enc.half().float(), mask, t, s, bits[s])
        worst = max(worst, rel_l2(out, exp))
        sched.per_step_callback(s, 0)
    sched.reset_step()
    return worst


def test_registered_plugins_run_on_native_sublayers_and_match_the_oracle():
    from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry, ComputeFFRegistry

    calls = []

    @ComputeFFRegistry.register
    def double_cached_ff(block, norm_hidden_states, **kwargs):  # the VERDICT's example: 2x the cached FF
        calls.append(("ff", block.block_num))
        return 2 * ComputeFFRegistry.default()(block, norm_hidden_states)

    @ComputeFFRegistry.register
    def scaled_fresh_ff(block, norm_hidden_states, scale=1.0):  # kwargs from the schedule JSON; never caches
        return scale * block.ff(norm_hidden_states)

    @ComputeAttnRegistry.register
    def attn_always_fresh(block, attn, hidden_states, encoder_hidden_states, attention_mask, **kwargs):  # ignores the flags: TGATE-style "never cache this one"
        calls.append((attn, block.block_num))
        return getattr(block, attn)(hidden_states, encoder_hidden_states, attention_mask)

    layers, steps = 3, 5
    bits = np.random.default_rng(4).random((steps, layers, 3)) < 0.5
    custom = {}
    for s in range(steps):
        custom[(s, 0)] = {"custom_compute_ff": {"name": "double_cached_ff"}}
        custom[(s, 2)] = {"custom_compute_attn": {"name": "attn_always_fresh", "kwargs": {}}}
        if s >= 2:  # block 1 names a plug-in at some steps only: the default callable serves the others, with the Python-side cache
            custom[(s, 1)] = {"custom_compute_ff": {"name": "scaled_fresh_ff", "kwargs": {"scale": 0.5}}}
    tr, sched, cfg, w = _model(layers, steps, bits, custom)
    assert tr.uses_plugins and sched.custom_component_mask().tolist() == [[0, 0, 1], [0, 0, 1], [1, 1, 0]]
    ref = PixArtOracle(cfg, w)
    ref.plugins = {(0, "ff"): lambda cached, fresh: 2 * cached(), (2, "attn1"): lambda cached, fresh: fresh(), (2, "attn2"): lambda cached, fresh: fresh()}
    # block 1 ff: default at steps 0-1, 0.5 * fresh from step 2 on -- the oracle switches with the step through a mutable cell
    step_cell = {"s": 0}
    ref.plugins[(1, "ff")] = lambda cached, fresh: cached() if step_cell["s"] < 2 else 0.5 * fresh()
    orig_forward = ref.forward

    def fwd(*a, **k):
        step_cell["s"] = a[4]
        return orig_forward(*a, **k)

    ref.forward = fwd
    worst = _run(tr, sched, ref, bits, steps)
    print(f"plug-in schedule worst per-step rel-L2: {worst:.3e}")
    assert within(worst, 2e-3)
    assert ("ff", "0") in calls and ("attn1", "2") in calls and ("attn2", "2") in calls and len(calls) == steps * 3
    # the whole-loop graph cannot call back into Python: refused, not silently run without the plug-ins
    inp = synthetic_inputs(2)
    with pytest.raises(ValueError, match="plug-in"):
        tr.engine.run_schedule(torch.zeros(4, 120, 4096), None, inp["latents"], np.zeros(steps, np.float32), np.zeros((steps, 5), np.float32))

    # an exception inside a plug-in surfaces as itself
    @ComputeFFRegistry.register
    def broken_ff(block, norm_hidden_states):
        raise KeyError("boom")

    tr2, sched2, _, _ = _model(2, 2, np.ones((2, 2, 3), bool), {(0, 1): {"custom_compute_ff": {"name": "broken_ff"}}})
    with pytest.raises(KeyError, match="boom"):
        tr2(torch.zeros(2, 4, 32, 32).cuda(), encoder_hidden_states=torch.zeros(2, 120, 4096).cuda(), timestep=torch.zeros(2).cuda())
    # a name nobody registered is an error when the schedule is attached (the reference asserts at the first call)
    with pytest.raises(ValueError, match="not found in registry"):
        _model(2, 2, np.ones((2, 2, 3), bool), {(1, 0): {"custom_compute_attn": {"name": "no_such_function"}}})


def test_defaults_are_real_callables_on_a_block():
    """compute_attn_cached / compute_ff_cached are no longer markers: called on a block of the facade they run the native sub-layers and keep the cache
    attribute, as the reference's do (cached_transformer_block.py:326-391)."""
    from ecad_amd.transformer_blocks.custom_attn_ff import ComputeAttnRegistry, ComputeFFRegistry

    tr, sched, cfg, w = _model(2, 2, np.array([[[1, 1, 1]] * 2, [[0, 0, 0]] * 2], bool), {})
    assert not tr.uses_plugins
    ref = PixArtOracle(cfg, w)
    inp = synthetic_inputs(1)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0)
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0)
    tr.engine.begin_image_batch(enc.cuda(), mask.cuda(), 32, 32)
    blk = tr.transformer_blocks[1]
    x = torch.randn(2, 256, 1152, generator=torch.Generator().manual_seed(1)).half()
    ff = ComputeFFRegistry.get(None)(blk, x.cuda())
    assert within(rel_l2(ff, ref.feed_forward("transformer_blocks.1.ff", x.float())), 9e-4) and blk.cached_ff_output is ff
    sched.per_step_callback(0, 0)  # step 1: flag false -> the cached tensor comes back, nothing is launched
    assert ComputeFFRegistry.get("compute_ff_cached")(blk, torch.zeros_like(x).cuda()) is ff
    sched.reset_step()
    a1 = ComputeAttnRegistry.get(None)(blk, "attn1", x.cuda(), None, None)
    assert within(rel_l2(a1, ref.attention("transformer_blocks.1.attn1", x.float(), x.float(), None)), 9e-4)
    ctx, bias = tr.batch_context()
    a2 = blk.compute_attn("attn2", x.cuda(), ctx, bias)
    octx = ref.caption_projection(ref.q(enc.half().float()))
    obias = ((1 - mask.float()) * -10000.0)[:, None, :]
    assert within(rel_l2(ctx, octx), 8e-4)
    assert within(rel_l2(a2, ref.attention("transformer_blocks.1.attn2", x.float(), octx, obias)), 8e-4)
    with pytest.raises(ValueError, match="Invalid attention type"):
        ComputeAttnRegistry.get(None)(blk, "attn3", x.cuda(), None, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ComputeFFRegistry.get(None)(object(), x)


@pytest.mark.parametrize("reset_before", [1, 3])
def test_reset_cache_mid_run_recomputes_like_the_reference(reset_before):
    """reset_cache between two steps (pixart_transformer_2d_edited.py:155-158): the next forward finds every cache empty and recomputes all components whatever
    its flags say (cached_transformer_block.py:340-347), later steps follow their flags again.  The engine lowers that step ad hoc from its live cache state."""
    layers, steps = 3, 6
    bits = np.random.default_rng(8).random((steps, layers, 3)) < 0.4
    bits[0] = True
    tr, sched, cfg, w = _model(layers, steps, bits, {})
    ref = PixArtOracle(cfg, w)
    worst = _run(tr, sched, ref, bits, steps, reset_before=reset_before)
    print(f"reset before step {reset_before}: worst per-step rel-L2 {worst:.3e}")
    assert within(worst, 2e-3)
    st = tr.engine.stats()
    assert st["dynamic_step_plans"] >= 1
    executed_after_reset = [e for e in ref.cache.executed_log if e[0] == reset_before]
    assert len(executed_after_reset) == layers * 3  # the oracle recomputed everything at that step too


def test_plan_sets_and_graphs_are_cached_by_schedule_content():
    """Alternating between a few schedules (the GA's uncached reference run + candidates; SURVEY 7-4 'graph cache keyed by schedule hash') re-uses the
    lowered op lists and the instantiated hipGraph of each: no new lowering, no new capture, bit-identical results."""
    from ecad_amd.engine import EcadEngine
    from oracle.dpm_solver_ref import DPMSolverPP2M

    layers, steps, B = 2, 4, 1
    cfg = PixArtConfig(num_layers=layers)
    eng = EcadEngine(num_layers=layers, max_batch=2 * B, max_tokens=256, max_text_tokens=120)
    eng.load_state_dict(make_weights(cfg, seed=1))
    inp = synthetic_inputs(B)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0).cuda()
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0).cuda()
    solver = DPMSolverPP2M(steps)
    coefs = np.array([solver.coefficients(i, solver.order_for_step(i)) for i in range(steps)], dtype=np.float32)
    rng = np.random.default_rng(0)
    scheds = [np.ones((steps, layers, 3), bool)] + [rng.random((steps, layers, 3)) < 0.5 for _ in range(2)]
    first = []
    for b in scheds:
        eng.set_schedule(b)
        first.append(eng.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True).cpu())
    st0 = eng.stats()
    assert st0["plan_builds"] == 3 and st0["graph_captures"] == 3
    for _ in range(3):
        for b, want in zip(scheds, first):
            eng.set_schedule(b)
            assert torch.equal(eng.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True).cpu(), want)
    st1 = eng.stats()
    assert st1["plan_builds"] == 3 and st1["graph_captures"] == 3 and st1["plan_sets_cached"] == 2
    # more distinct schedules than the cache holds (4 parked + 1 live): the oldest is evicted and rebuilt on return
    for k in range(6):
        eng.set_schedule(rng.random((steps, layers, 3)) < 0.5)
        eng.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True)
    eng.set_schedule(scheds[1])
    assert torch.equal(eng.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True).cpu(), first[1])
    st2 = eng.stats()
    assert st2["plan_builds"] == 3 + 6 + 1 and st2["plan_sets_cached"] == 4
    eng.close()


def test_self_attention_mask_and_encoder_bias_as_the_reference_takes_them():
    """forward(attention_mask=..., encoder_attention_mask=...) (reference pixart_transformer_2d_edited.py:166-167,213-215,255-291): a 2-D self-attention mask
    becomes the bias (1 - m) * -10000 over the image keys, and a 3-D encoder_attention_mask IS a bias and is added as given -- here one with values that are
    neither 0 nor -10000, which a binarising shim gets wrong.  Engine (ecad_set_attention_bias) against the oracle with the same arguments."""
    layers, steps, B = 2, 3, 2
    bits = np.ones((steps, layers, 3), dtype=bool)
    bits[1, 0, 0] = bits[1, 1, 1] = bits[2, 0, 2] = False
    tr, sched, cfg, w = _model(layers, steps, bits, {})
    ref = PixArtOracle(cfg, w)
    inp = synthetic_inputs(B)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0)
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0)
    g = torch.Generator().manual_seed(21)
    self_mask = (torch.rand(2 * B, 256, generator=g) > 0.3).to(torch.int64)  # 2-D: 1 keep, 0 discard
    self_mask[:, 0] = 1
    enc_bias = ((1 - mask.float()) * -10000.0 + torch.randn(2 * B, 120, generator=g) * 0.5)[:, None, :].half()  # 3-D: a bias, soft values on the kept keys
    worst = 0.0
    for s in range(steps):
        x = torch.randn(2 * B, 4, 32, 32, generator=g)
        t = torch.full((2 * B,), 900.0 - 70 * s)
        out = tr(x.cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=enc_bias.cuda(), attention_mask=self_mask.cuda(), timestep=t.cuda(), return_dict=False)[0]
        exp = ref.forward(x.half().float(), enc.half().float(), enc_bias.float(), t, s, bits[s], attention_mask=self_mask)
        worst = max(worst, rel_l2(out, exp))
        sched.per_step_callback(s, 0)
    sched.reset_step()
    assert within(worst, 2e-3)
    # the masks matter: the same inputs without them give a different prediction (so a shim that dropped them could not pass)
    tr.reset_cache()
    out_plain = tr(x.cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=mask.cuda(), timestep=t.cuda(), return_dict=False)[0]
    ref2 = PixArtOracle(cfg, w)
    exp_plain = ref2.forward(x.half().float(), enc.half().float(), mask, t, 0, bits[0])
    assert within(rel_l2(out_plain, exp_plain), 2e-3)
    assert rel_l2(out_plain, exp) > 1e-2
    with pytest.raises(ValueError, match="varies with the query row"):
        tr(x.cuda(), encoder_hidden_states=enc.cuda(), encoder_attention_mask=mask.cuda(), attention_mask=torch.zeros(2 * B, 256, 256).cuda(), timestep=t.cuda())


def test_a_second_forward_at_step_zero_reads_the_warm_caches_like_the_reference():
    """The reference drops its caches only in the callback after the last step (image_generator.py:193-202): a harness that calls forward twice at step 0 with
    the same inputs and fires no callback (warm-up + measure) gets, the second time, the cached sub-layer outputs wherever the schedule's flag is false."""
    layers, steps, B = 3, 2, 1
    bits = np.ones((steps, layers, 3), dtype=bool)
    bits[0, 0, 2] = bits[0, 1, 0] = bits[0, 2, 1] = False  # step 0: forced recompute on an empty cache, replay on a warm one
    tr, sched, cfg, w = _model(layers, steps, bits, {})
    ref = PixArtOracle(cfg, w)
    inp = synthetic_inputs(B)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0).cuda()
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0).cuda()
    g = torch.Generator().manual_seed(5)
    x1, x2 = torch.randn(2 * B, 4, 32, 32, generator=g), torch.randn(2 * B, 4, 32, 32, generator=g)
    t = torch.full((2 * B,), 999.0)
    o1 = tr(x1.cuda(), encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=t.cuda(), return_dict=False)[0]
    o2 = tr(x2.cuda(), encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=t.cuda(), return_dict=False)[0]  # same step, no callback in between
    e1 = ref.forward(x1.half().float(), enc.cpu().half().float(), mask.cpu(), t, 0, bits[0])
    e2 = ref.forward(x2.half().float(), enc.cpu().half().float(), mask.cpu(), t, 0, bits[0])  # the oracle's caches are warm too
    assert within(rel_l2(o1, e1), 2e-3) and within(rel_l2(o2, e2), 2e-3)
    cold = PixArtOracle(cfg, w).forward(x2.half().float(), enc.cpu().half().float(), mask.cpu(), t, 0, bits[0])
    assert rel_l2(e2, cold) > 5e-3  # warm and cold differ by far more than the tolerance (measured 8.8e-3): the engine really replayed
    assert tr.engine.stats()["dynamic_step_plans"] >= 1
    tr.reset_cache()  # what the last-step callback does: now step 0 is cold again
    o3 = tr(x2.cuda(), encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=t.cuda(), return_dict=False)[0]
    assert within(rel_l2(o3, cold), 2e-3)


def test_blocks_run_in_the_order_of_the_references_default_fx_graph(golden_dir):
    """SURVEY 8(c) fixture 5: the node order of the reference's default DiT graph (captured by tests/golden/make_golden.py from graph/pixart_builder.py:96-124,
    187-204) is transformer_block_0 .. 27, each fed by its predecessor.  The engine's order is observed through the plug-in callbacks of a 28-block model."""
    import json

    gold = json.loads((golden_dir / "dit_graph_vectors.json").read_text())
    order = [int(n.rsplit("_", 1)[1]) for n in gold["steps"]["000"]["call_module_order"]]
    layers, steps = gold["num_blocks"], 2
    from ecad_amd.transformer_blocks.custom_attn_ff import ComputeFFRegistry

    seen = []

    @ComputeFFRegistry.register
    def order_spy(block, norm_hidden, **kw):
        seen.append(int(block.block_num))
        return block.ff(norm_hidden)

    bits = np.ones((steps, layers, 3), dtype=bool)
    custom = {(s, b): {"custom_compute_ff": {"name": "order_spy", "kwargs": {}}} for s in range(steps) for b in range(layers)}
    tr, sched, cfg, w = _model(layers, steps, bits, custom)
    inp = synthetic_inputs(1)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0).cuda()
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0).cuda()
    x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(1))
    for s in range(steps):
        tr(x.cuda(), encoder_hidden_states=enc, encoder_attention_mask=mask, timestep=torch.full((2,), 900.0 - 100 * s).cuda(), return_dict=False)
        sched.per_step_callback(s, 0)
    assert seen == order * steps


def test_native_variants_run_per_step_and_inside_the_whole_loop_graph():
    """Round 6 (row a6): a registered compute function may declare a native variant -- a gate rule the engine lowers into its own kernels
    (``ecad_set_native_variants``) -- so a schedule that names it stays on the device: per step through ``ecad_step`` and inside the whole-loop plan and its hipGraph
    (``ecad_run_schedule``), which a host callback cannot enter.  ``compute_attn_recompute`` / ``compute_ff_recompute`` ignore the schedule's flag;
    ``compute_attn_cfg_mean_cache`` caches the mean of the CFG halves of attn2 while the hidden states take the un-averaged output (the caching rule of
    ``compute_attn_tgate``, cached_transformer_block.py:443-452).  Checked against the oracle with the same rules as hooks: per step (2e-3, as every per-step
    comparison here), on the final latents of the whole loop (6e-3, test_gpu_engine's bound for final latents), graph == eager bit for bit, and the executed-component
    count of EcadStats follows the variants."""
    from oracle.dpm_solver_ref import DPMSolverPP2M
    from oracle.pipeline_ref import denoise

    layers, steps = 3, 5
    bits = np.random.default_rng(14).random((steps, layers, 3)) < 0.5
    bits[0] = True
    custom = {}
    for s in range(steps):
        if s >= 1:
            custom[(s, 0)] = {"custom_compute_attn": {"name": "compute_attn_recompute"}}         # block 0: attn1 and attn2 always evaluated from step 1 on
        if s >= 2:
            custom.setdefault((s, 1), {})["custom_compute_ff"] = {"name": "compute_ff_recompute"}    # block 1: ff always evaluated from step 2 on
        custom.setdefault((s, 2), {})["custom_compute_attn"] = {"name": "compute_attn_cfg_mean_cache", "kwargs": {}}   # block 2: attn2 caches the CFG mean at every step
    tr, sched, cfg, w = _model(layers, steps, bits, custom, unit_gates=True)
    assert not tr.uses_plugins and tr.uses_native_variants and sched.uses_only_native_compute() and not sched.custom_component_mask().any()
    kinds = sched.native_variant_kinds()
    assert kinds[0, 0].tolist() == [0, 0, 0] and kinds[1, 0].tolist() == [1, 1, 0] and kinds[1, 1].tolist() == [0, 0, 0] and kinds[2, 1].tolist() == [0, 0, 1]
    assert all(kinds[s, 2].tolist() == [0, 2, 0] for s in range(steps))   # attn1 of the CFG-mean function is the default rule
    forced = int(((kinds == 1) & ~bits).sum())
    ex = np.ones((layers, 3), bool)
    expected_executed = 0
    have = np.zeros((layers, 3), bool)
    for s in range(steps):
        ex = bits[s] | (kinds[s] == 1) | ~have
        have |= ex
        expected_executed += int(ex.sum())
    assert tr.engine.stats()["components_executed"] == expected_executed and forced > 0

    def make_ref():
        ref = PixArtOracle(cfg, w)
        cell = {"s": 0}

        def fresh_and_cache(block, comp, first):
            def hook(cached, fresh):
                if cell["s"] < first:
                    return cached()
                out = fresh()
                ref.cache.entries[(block, comp)] = out
                return out
            return hook

        def cfg_mean(cached, fresh):
            n0 = len(ref.cache.executed_log)
            out = cached()
            if len(ref.cache.executed_log) > n0:   # evaluated at this step: the cache keeps the CFG mean, the block adds the un-averaged output
                u, t = out.chunk(2)
                m = ref.q(0.5 * (u + t))
                ref.cache.entries[(2, "attn2")] = torch.cat([m, m])
            return out

        ref.plugins = {(0, "attn1"): fresh_and_cache(0, "attn1", 1), (0, "attn2"): fresh_and_cache(0, "attn2", 1), (1, "ff"): fresh_and_cache(1, "ff", 2), (2, "attn2"): cfg_mean}
        orig = ref.forward

        def fwd(*a, **k):
            cell["s"] = a[4]
            return orig(*a, **k)

        ref.forward = fwd
        return ref

    worst = _run(tr, sched, make_ref(), bits, steps)
    print(f"native-variant schedule, per step: worst rel-L2 {worst:.3e}")
    assert within(worst, 2e-3)
    # the whole loop: one hipGraph, no host round trips -- and the same values as launching its ops eagerly
    B = 2
    inp = synthetic_inputs(B)
    enc = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]], 0).cuda()
    mask = torch.cat([inp["negative_prompt_attention_mask"], inp["prompt_attention_mask"]], 0).cuda()
    solver = DPMSolverPP2M(steps)
    coefs = np.array([solver.coefficients(i, solver.order_for_step(i)) for i in range(steps)], dtype=np.float32)
    graph = tr.engine.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True).clone()
    assert tr.engine.stats()["graph_nodes"] > 0
    eager = tr.engine.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=False)
    assert torch.equal(graph, eager)
    q = {k: (v.half().float() if v.is_floating_point() else v) for k, v in inp.items()}
    exp = denoise(make_ref(), bits, q)
    err = rel_l2(graph, exp)
    print(f"native-variant schedule, whole loop in one hipGraph: final-latent rel-L2 {err:.3e}")
    assert within(err, 6e-3)
    # the rule is part of the result: the same bits without the variants give other latents (the net would notice a variant that is silently dropped)
    tr.engine.set_native_variants(None)
    plain = tr.engine.run_schedule(enc, mask, inp["latents"].cuda(), solver.timesteps, coefs, 4.5, use_graph=True)
    moved = rel_l2(plain, exp)
    print(f"the same schedule bits without the variants: {moved:.3e} from the oracle that applies them")
    assert moved > 3 * err
    # limits are named: the CFG-mean rule exists for attn2, ids are checked, the geometry must be the schedule's
    bad = np.zeros((steps, layers, 3), np.uint8)
    bad[0, 0, 0] = 2
    with pytest.raises((RuntimeError, ValueError), match="attn2"):
        tr.engine.set_native_variants(bad)
    bad[0, 0, 0] = 9
    with pytest.raises(ValueError, match="unknown"):
        tr.engine.set_native_variants(bad)
    with pytest.raises(ValueError, match="schedule"):
        tr.engine.set_native_variants(np.ones((steps + 1, layers, 3), np.uint8))
