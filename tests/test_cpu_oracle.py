"""CPU suite (-m "not gpu"): the oracle against the reference's golden vectors, host logic, ABI surface."""
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import schedule_ref as S
from oracle.dpm_solver_ref import DPMSolverPP2M
from oracle.pixart_ref import PixArtConfig, PixArtOracle, make_weights, sincos_pos_embed_2d, weight_shapes

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"


# ---------------------------------------------------------------------------------------------
# schedule semantics pinned on the reference's own data
# ---------------------------------------------------------------------------------------------
def test_mac_kat_every_reference_schedule():
    """Every per-step MAC count the reference measured (calflops, compute_macs.py:296-303) in its 1090
    PixArt-256 schedule JSONs is reproduced by gate rule + closed-form cost model."""
    kat = np.load(GOLD / "mac_kat.npz")
    assert len(kat["names"]) == 1090
    bad = []
    for name, packed, macs, total in zip(kat["names"], kat["bits"], kat["macs"], kat["totals"]):
        bits = np.unpackbits(packed)[:1680].reshape(20, 28, 3).astype(bool)
        got = S.pixart_mac_model(S.resolve_executed(bits))
        if not np.array_equal(got, macs) or (total >= 0 and got.sum() != total):
            bad.append(str(name))
    assert not bad, bad[:5]


def test_ours_fast_known_answers():
    bits = S.bits_from_schedule_json(ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json")
    vec = json.load(open(GOLD / "schedule_vectors.json"))["to_numpy"]["ours_fast"]
    assert np.packbits(bits.flatten()).tobytes().hex() == vec["packed_hex"]
    assert bits.sum() == vec["sum"] == 598
    ex = S.resolve_executed(bits)
    assert ex.sum() == 599 and ex[0, 17, 0] and not bits[0, 17, 0]  # the one forced recompute
    macs = S.pixart_mac_model(ex)
    assert macs[0] == 285689806848 and macs[1] == 9652174848 and macs[3] == 1498447872
    assert macs.sum() == 2134989471744
    assert S.attention_core_macs(ex) == 89389006848
    pure_hit = [s for s in range(20) if not ex[s].any()]
    assert pure_hit == [3, 5, 7, 8, 11, 13]


def test_step_counter_matches_reference_trace():
    vec = json.load(open(GOLD / "schedule_vectors.json"))["step_trace"]
    bits = S.bits_from_schedule_json(ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json")
    c = S.StepCounter(20)
    for s, (curr, ff3, attn0) in enumerate(vec["trace"]):
        assert c.curr_step == curr == s
        assert bool(bits[c.curr_step, 3, 2]) == ff3 and bool(bits[c.curr_step, 0, 0]) == attn0
        if s < 19:
            c.per_step_callback(s)
    assert vec["curr_step_after_19"] == 20  # the schedule object alone just counts
    c.per_step_callback(19)  # with the reset callback last in the chain the counter returns to 0
    assert c.curr_step == 0 == vec["curr_step_after_reset"] and c.resets == 1


# ---------------------------------------------------------------------------------------------
# oracle arithmetic: internal invariants (diffusers itself is not importable: parity unpinned)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny():
    cfg = PixArtConfig(num_layers=2, num_attention_heads=2, attention_head_dim=8, caption_channels=32)
    return cfg, make_weights(cfg, seed=1)


def _inputs(cfg, B=2, T=7):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 4, 32, 32, generator=g)
    enc = torch.randn(B, T, cfg.caption_channels, generator=g)
    mask = torch.ones(B, T, dtype=torch.int64)
    mask[1, 4:] = 0
    return x, enc, mask


def test_all_true_schedule_equals_uncached_model(tiny):
    cfg, w = tiny
    x, enc, mask = _inputs(cfg)
    t = torch.tensor([500.0, 500.0])
    a, b = PixArtOracle(cfg, w), PixArtOracle(cfg, w)
    flags = np.ones((cfg.num_layers, 3), bool)
    o1 = a.forward(x, enc, mask, t, 0, flags)
    o1b = a.forward(x, enc, mask, t, 1, flags)  # caches filled, all recompute: same answer
    b.reset_cache()
    o2 = b.forward(x, enc, mask, t, 0, flags)
    assert torch.equal(o1, o2) and torch.equal(o1, o1b)


def test_cached_step_is_gated_sum(tiny):
    """fully cached step == h += g_msa*C1 + C2 + g_mlp*C3 per block, gates from the CURRENT timestep."""
    cfg, w = tiny
    x, enc, mask = _inputs(cfg)
    m = PixArtOracle(cfg, w)
    ones, zeros = np.ones((cfg.num_layers, 3), bool), np.zeros((cfg.num_layers, 3), bool)
    m.forward(x, enc, mask, torch.tensor([800.0, 800.0]), 0, ones)
    t1 = torch.tensor([300.0, 300.0])
    x1 = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(9))
    out = m.forward(x1, enc, mask, t1, 1, zeros)
    h = m.patch_embed(x1)
    mod, emb = m.adaln_single(t1)
    for i in range(cfg.num_layers):
        tbl = m.w[f"transformer_blocks.{i}.scale_shift_table"][None] + mod.reshape(2, 6, -1)
        h = h + tbl[:, 2:3] * m.cache.entries[(i, "attn1")]
        h = h + m.cache.entries[(i, "attn2")]
        h = h + tbl[:, 5:6] * m.cache.entries[(i, "ff")]
    assert torch.allclose(out, m.create_output(h, emb), atol=1e-5, rtol=1e-5)


def test_masked_text_tokens_do_not_matter(tiny):
    cfg, w = tiny
    x, enc, mask = _inputs(cfg)
    t = torch.tensor([100.0, 100.0])
    flags = np.ones((cfg.num_layers, 3), bool)
    a = PixArtOracle(cfg, w).forward(x, enc, mask, t, 0, flags)
    enc2 = enc.clone()
    enc2[1, 4:] += 3.0  # discarded tokens (bias -10000)
    b = PixArtOracle(cfg, w).forward(x, enc2, mask, t, 0, flags)
    assert torch.allclose(a, b, atol=1e-5)


def test_oracle_executes_exactly_the_components_the_pinned_rule_predicts():
    """The gate of the oracle's module graph (PixArtOracle._gated_default, restating cached_transformer_block.py:340-360) tied to the rule pinned on the
    reference's 1090 measured schedules (schedule_ref.resolve_executed): on `ours_fast` (reduced width, 28 blocks, 20 steps) the SET of (step, block,
    component) evaluations is the same, not just their count."""
    bits = S.bits_from_schedule_json(ROOT / "schedules/schedules_in_paper/pixart_alpha_256/ours_fast.json")
    cfg = PixArtConfig(num_layers=28, num_attention_heads=2, attention_head_dim=8, caption_channels=32, sample_size=8)
    m = PixArtOracle(cfg, make_weights(cfg, seed=2))
    g = torch.Generator().manual_seed(0)
    enc, mask = torch.randn(2, 5, cfg.caption_channels, generator=g), torch.ones(2, 5, dtype=torch.int64)
    for s in range(bits.shape[0]):
        m.forward(torch.randn(2, 4, 8, 8, generator=g), enc, mask, torch.full((2,), 999.0 - 50 * s), s, bits[s])
    comp = {"attn1": 0, "attn2": 1, "ff": 2}
    ran = {(s, b, comp[c]) for s, b, c in m.cache.executed_log}
    want = {tuple(int(v) for v in idx) for idx in np.argwhere(S.resolve_executed(bits))}
    assert len(m.cache.executed_log) == len(ran) == 599
    assert ran == want
    assert (0, 17, 0) in ran and not bits[0, 17, 0]  # the forced recompute inside ours_fast itself (flag false, cache empty)


def test_default_dit_graph_of_the_reference_is_the_sequential_block_loop():
    """SURVEY 8(c) fixture 5, captured by tests/golden/make_golden.py from the reference's own builder (graph/pixart_builder.py:96-124,187-204): the fx graph of
    the default DiT schedule calls transformer_block_0 .. 27 in order, each on its predecessor's output, at every step; on order-sensitive stand-in blocks it
    returns what `for block in blocks` returns.  That loop is what the oracle and the engine hard-wire."""
    gold = json.load(open(GOLD / "dit_graph_vectors.json"))
    n = gold["num_blocks"]
    assert n == 28 and gold["name"] == "default" and set(gold["steps"]) == {"000", "007", "019"}
    for step in gold["steps"].values():
        assert step["call_module_order"] == [f"transformer_block_{i}" for i in range(n)]
        assert step["equals_sequential_loop"] is True
        wiring = step["hidden_states_input_of"]
        assert wiring["_0_0"] == "_0_0_input_func"
        feeds = {node[2]: node for node in step["nodes"]}
        for i in range(1, n):  # block i reads an identity of block i - 1's output and nothing else
            assert wiring[f"_{i}_0"] == f"_{i}_0_input_func" and feeds[f"_{i}_0_input_func"][1] == "identity"
    x = torch.tensor(gold["x"])
    for a, b in zip(gold["stub_scale"], gold["stub_shift"]):
        x = x * a + b
    assert torch.equal(x, torch.tensor(gold["sequential_out"]))
    # the oracle visits the blocks in that order
    cfg = PixArtConfig(num_layers=n, num_attention_heads=2, attention_head_dim=8, caption_channels=32, sample_size=8)
    m = PixArtOracle(cfg, make_weights(cfg, seed=2))
    g = torch.Generator().manual_seed(0)
    m.forward(torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 5, 32, generator=g), torch.ones(1, 5, dtype=torch.int64), torch.tensor([500.0]), 0, np.ones((n, 3), bool))
    assert [b for _, b, c in m.cache.executed_log if c == "ff"] == list(range(n))


def test_attention_masks_as_the_reference_converts_them(tiny):
    """_create_attention_mask (pixart_transformer_2d_edited.py:255-291): a 2-D mask and the 3-D bias (1 - m) * -10000 built from it are the same thing, for the
    self-attention mask and for the encoder mask; a soft 3-D bias is added as given (it changes the result, a binarised one would not)."""
    cfg, w = tiny
    x, enc, mask = _inputs(cfg)
    t = torch.tensor([100.0, 100.0])
    flags = np.ones((cfg.num_layers, 3), bool)
    g = torch.Generator().manual_seed(3)
    self_mask = (torch.rand(2, 256, generator=g) > 0.4).to(torch.int64)
    self_mask[:, 0] = 1
    a = PixArtOracle(cfg, w).forward(x, enc, mask, t, 0, flags, attention_mask=self_mask)
    b = PixArtOracle(cfg, w).forward(x, enc, ((1 - mask.float()) * -10000.0)[:, None, :], t, 0, flags, attention_mask=((1 - self_mask.float()) * -10000.0)[:, None, :])
    plain = PixArtOracle(cfg, w).forward(x, enc, mask, t, 0, flags)
    assert torch.equal(a, b) and float((a - plain).abs().max()) > 1e-5  # (a 2-block toy with N(0, 0.02) weights: small, but far above rounding)
    soft = ((1 - mask.float()) * -10000.0 + torch.randn(2, 7, generator=g))[:, None, :]
    c = PixArtOracle(cfg, w).forward(x, enc, soft, t, 0, flags)
    assert float((c - plain).abs().max()) > 1e-6


def test_pos_embed_table():
    tab = sincos_pos_embed_2d(1152, 16, base_size=16)
    assert tab.shape == (256, 1152)
    assert torch.allclose(tab[0, :288], torch.zeros(288)) and torch.allclose(tab[0, 288:576], torch.ones(288))  # sin 0 / cos 0
    assert torch.allclose(tab[1, 0], torch.sin(torch.tensor(1.0)))  # token 1 = row 0, col 1: first half encodes the column
    assert torch.allclose(tab[16, 576], torch.sin(torch.tensor(1.0)))  # token 16 = row 1, col 0: second half encodes the row
    assert torch.allclose(tab[16, :288], torch.zeros(288))


def test_dpm_solver_timesteps_and_last_step():
    s = DPMSolverPP2M(20)
    assert s.timesteps.tolist() == [999, 949, 899, 849, 799, 749, 699, 649, 599, 549, 500, 450, 400, 350, 300, 250, 200, 150, 100, 50]
    assert s.sigmas[-1] == 0 and len(s.sigmas) == 21
    a_s, sg_s, c_x, c0, c1 = s.coefficients(19, 1)
    assert c_x == 0 and abs(c0 - 1) < 1e-12 and c1 == 0  # final sigma 0: x <- x0
    # a constant-eps trajectory: second-order and first-order updates agree when x0 does not change
    _, _, cx1, a1, _ = s.coefficients(5, 1)
    _, _, cx2, b0, b1 = s.coefficients(5, 2)
    assert abs(cx1 - cx2) < 1e-12 and abs(a1 - (b0 + b1)) < 1e-9


def test_weight_shapes_match_pixart_xl2():
    n = sum(int(np.prod(s)) for s in weight_shapes(PixArtConfig()).values())
    assert 600e6 < n < 615e6  # PixArt-XL/2 ~0.6B parameters


# ---------------------------------------------------------------------------------------------
# micro-conditioning of the 1024-MS alpha checkpoint (use_additional_conditions)
# ---------------------------------------------------------------------------------------------
def test_alpha_1024_ms_mac_kat_pins_the_size_embedders():
    """The reference's measured per-step MACs for PixArt-XL-2-1024-MS (N = 4096, resolution + aspect-ratio embedders evaluated at every
    step) = the closed-form model with the MACs of exactly the extra Linear layers the oracle adds for use_additional_conditions."""
    vec = json.loads((GOLD / "alpha1024_vectors.json").read_text())
    assert vec
    extra = weight_shapes(PixArtConfig(use_additional_conditions=True)).keys() - weight_shapes(PixArtConfig()).keys()
    shapes = weight_shapes(PixArtConfig(use_additional_conditions=True))
    # resolution embedder runs on two scalars (height, width), aspect-ratio embedder on one
    size_macs = sum(int(np.prod(shapes[k])) * (2 if "resolution" in k else 1) for k in extra if k.endswith("weight"))
    assert size_macs == 737_280
    for name, v in vec.items():
        bits = np.unpackbits(np.frombuffer(bytes.fromhex(v["bits_packed_hex"]), dtype=np.uint8))[: v["steps"] * 84].reshape(v["steps"], 28, 3).astype(bool)
        assert int(bits.sum()) == v["bits_sum"]
        got = S.pixart_mac_model(S.resolve_executed(bits), tokens=(v["height"] // 16) * (v["width"] // 16), size_embedder_macs=size_macs)
        assert got.tolist() == v["macs"] and int(got.sum()) == v["total_macs"], name


def test_added_cond_kwargs_enter_the_conditioning():
    cfg = PixArtConfig(num_layers=1, num_attention_heads=3, attention_head_dim=8, caption_channels=32, use_additional_conditions=True)
    w = make_weights(cfg, seed=2)
    # the extra keys are appended: the shared keys draw the same tensors as without them
    w0 = make_weights(PixArtConfig(num_layers=1, num_attention_heads=3, attention_head_dim=8, caption_channels=32), seed=2)
    assert all(torch.equal(w[k], v) for k, v in w0.items())
    x, enc, mask = _inputs(cfg)
    t = torch.tensor([500.0, 500.0])
    flags = np.ones((1, 3), bool)
    m = PixArtOracle(cfg, w)
    with pytest.raises(ValueError, match="added_cond_kwargs"):
        m.forward(x, enc, mask, t, 0, flags)
    cond = {"resolution": torch.tensor([[1024.0, 1024.0], [512.0, 2048.0]]), "aspect_ratio": torch.tensor([[1.0], [0.25]])}
    out = m.forward(x, enc, mask, t, 0, flags, added_cond=cond)
    swapped = {"resolution": cond["resolution"].flip(0), "aspect_ratio": cond["aspect_ratio"].flip(0)}
    out2 = PixArtOracle(cfg, w).forward(x, enc, mask, t, 0, flags, added_cond=swapped)
    assert not torch.allclose(out, out2)
    # sample 0 of `out` and sample 1 of a run with swapped conditions and swapped samples agree: conditions act per sample
    out3 = PixArtOracle(cfg, w).forward(x.flip(0), enc.flip(0), mask.flip(0), t, 0, flags, added_cond=swapped)
    assert torch.allclose(out, out3.flip(0), atol=1e-5)
    mod, emb = m.adaln_single(t, cond)
    assert mod.shape == (2, 6 * cfg.dim) and emb.shape == (2, cfg.dim)


def test_dpm_solver_solves_the_probability_flow_ode():
    """Independent of any source the restatement was written from: for data ~ N(0, c^2 I) the optimal noise prediction is known in closed
    form and the probability-flow ODE has the exact solution x_t = x_T * sqrt((a_t^2 c^2 + s_t^2) / (a_T^2 c^2 + s_T^2)), (a, s) = (alpha, sigma) of the
    VP schedule.  DPM-Solver++(2M) must (a) be exact for a point mass (c = 0: the data prediction is constant), (b) converge to the ODE
    solution with second order in the number of steps."""
    rng = np.random.default_rng(0)
    xT = rng.standard_normal(64)

    def run(steps, c):
        s = DPMSolverPP2M(steps)
        x = xT.copy()
        for i in range(steps):
            a, sg = DPMSolverPP2M._alpha_sigma(np.float64(s.sigmas[i]))
            eps = sg * x / (a * a * c * c + sg * sg)  # E[eps | x_t] for x0 ~ N(0, c^2)
            x = np.asarray(s.step(eps, x), dtype=np.float64)
        a_T, s_T = DPMSolverPP2M._alpha_sigma(np.float64(s.sigmas[0]))
        a_0, s_0 = DPMSolverPP2M._alpha_sigma(np.float64(s.sigmas[-1]))  # (1, 0): the final sigma is zero
        exact = xT * np.sqrt((a_0**2 * c * c + s_0**2) / (a_T**2 * c * c + s_T**2))
        return np.abs(x - exact).max() / max(np.abs(exact).max(), 1e-30), x

    err0, x = run(20, 0.0)
    assert np.abs(x).max() < 1e-12  # point mass at 0: every step lands on the data prediction's ray, the last one on the point
    e20, _ = run(20, 0.5)
    e160, _ = run(160, 0.5)
    # 8x the steps: a first-order method gains 8x, a second-order one 64x (measured: 4.6e-2 -> 3.0e-4, i.e. 156x; the integer timestep grid
    # and the first-order first / last steps make individual doublings uneven, so only the overall rate is asserted)
    assert e20 < 6e-2 and e160 < e20 / 30


def test_oracle_attention_is_torch_scaled_dot_product_attention(tiny):
    """The layer arithmetic of the reference lives in diffusers 0.30.3 (absent here: "restated, not verified against diffusers").  What can be pinned without it: the
    one non-trivial primitive its `AttnProcessor2_0` calls is torch's own `F.scaled_dot_product_attention(q, k, v, attn_mask=<additive bias>, dropout_p=0,
    is_causal=False)` on `[B, heads, tokens, head_dim]` views -- installed here.  The oracle spells the attention out (`softmax(q k^T / sqrt(d_h) + bias) v`); this
    test holds that spelling to torch's SDPA on the same projections, for the self-attention (no mask) and for the cross-attention with the reference's
    `(1 - mask) * -10000` bias broadcast over heads and queries (pixart_transformer_2d_edited.py:271-291), and the FLUX oracle's joint attention likewise."""
    import torch.nn.functional as F

    cfg, w = tiny
    m = PixArtOracle(cfg, w)
    g = torch.Generator().manual_seed(5)
    B, N, T, D, H, dh = 2, 12, 7, cfg.dim, cfg.num_attention_heads, cfg.attention_head_dim
    x = torch.randn(B, N, D, generator=g)
    ctx = torch.randn(B, T, D, generator=g)
    mask = torch.ones(B, T, dtype=torch.int64)
    mask[1, 3:] = 0
    bias = m._mask_to_bias(mask)  # [B, 1, T]: 0 keep / -10000 discard

    def sdpa(prefix, x_q, x_kv, attn_bias):
        heads = lambda t: t.view(B, -1, H, dh).transpose(1, 2)
        q, k, v = (heads(F.linear(s, w[f"{prefix}.{n}.weight"].float(), w[f"{prefix}.{n}.bias"].float())) for n, s in (("to_q", x_q), ("to_k", x_kv), ("to_v", x_kv)))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None if attn_bias is None else attn_bias[:, None], dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, H * dh)
        return F.linear(o, w[prefix + ".to_out.0.weight"].float(), w[prefix + ".to_out.0.bias"].float())

    p = "transformer_blocks.0"
    assert torch.allclose(m.attention(p + ".attn1", x, x, None), sdpa(p + ".attn1", x, x, None), atol=2e-6, rtol=1e-5)
    assert torch.allclose(m.attention(p + ".attn2", x, ctx, bias), sdpa(p + ".attn2", x, ctx, bias), atol=2e-6, rtol=1e-5)
    # a discarded key really is out: -10000 in front of the softmax leaves exp(-10000 + ...) == 0 in fp32
    ctx2 = ctx.clone()
    ctx2[1, 3:] += 5.0
    assert torch.allclose(m.attention(p + ".attn2", x, ctx2, bias)[1], m.attention(p + ".attn2", x, ctx, bias)[1], atol=1e-6)

    from oracle import flux_ref

    fm = flux_ref.FluxOracle(flux_ref.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16),
                             flux_ref.make_weights(flux_ref.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16), seed=2))
    dhf = fm.cfg.attention_head_dim
    qh, kh, vh = (torch.randn(2, 2, 9, dhf, generator=g) for _ in range(3))
    want = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=0.0, is_causal=False).transpose(1, 2).reshape(2, 9, 2 * dhf)
    assert torch.allclose(fm.sdpa(qh, kh, vh), want, atol=2e-6, rtol=1e-5)


def test_oracle_feed_forward_and_layer_norm_are_torch_modules(tiny):
    """Same idea for the other two layer kinds: `FeedForward(activation_fn="gelu-approximate")` of diffusers is Linear -> torch.nn.GELU(approximate="tanh") -> Linear,
    and `norm1 / norm2` are torch.nn.LayerNorm(D, elementwise_affine=False, eps=1e-6): both torch modules, instantiated here with the oracle's weights."""
    cfg, w = tiny
    m = PixArtOracle(cfg, w)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 12, cfg.dim, generator=g)
    p = "transformer_blocks.1.ff"
    l1, l2 = torch.nn.Linear(cfg.dim, 4 * cfg.dim), torch.nn.Linear(4 * cfg.dim, cfg.dim)
    with torch.no_grad():
        l1.weight.copy_(w[p + ".net.0.proj.weight"]); l1.bias.copy_(w[p + ".net.0.proj.bias"]); l2.weight.copy_(w[p + ".net.2.weight"]); l2.bias.copy_(w[p + ".net.2.bias"])
        want = l2(torch.nn.GELU(approximate="tanh")(l1(x)))
        ln = torch.nn.LayerNorm(cfg.dim, elementwise_affine=False, eps=1e-6)
        assert torch.allclose(m.feed_forward(p, x), want, atol=2e-6, rtol=1e-5)
        assert torch.allclose(m.layer_norm(x), ln(x), atol=1e-6, rtol=1e-6)
