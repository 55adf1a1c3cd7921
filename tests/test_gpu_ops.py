"""Per-kernel parity: HIP kernel (through the C ABI) vs a plain PyTorch fp32 reference of the same op.

Tolerances (floating point, stated per test): inputs are rounded to the storage type first, the
reference is evaluated in fp32 on those rounded values, so the remaining error is the kernel's
fp32-accumulate + one output rounding: rel-L2 <= 4.5e-4 for fp16, 3.6e-3 for bf16 (measured on MI355X: 2.1e-4 / 1.7e-3 = one output
rounding; attention 2.8e-4 / 2.2e-3 -> 6e-4 / 4.5e-3).  Every tolerance here is <= 2x the value `pytest -s` prints ([parity] lines).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from gpu_util import rel_l2, small_engine, within

pytestmark = pytest.mark.gpu

TOL = {"float16": 4.5e-4, "bfloat16": 3.6e-3}
ATT_TOL = {"float16": 6e-4, "bfloat16": 4.5e-3}


def _r(shape, dtype, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("M,N,K", [(512, 1152, 1152), (8192, 3456, 1152), (240, 2304, 1152), (20, 6912, 1152), (512, 1152, 4608), (130, 128, 256)])
def test_gemm_bias(dtype_name, M, N, K):
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    a, w = _r((M, K), dt, 1.0, 1), _r((N, K), dt, 0.05, 2)
    bias = _r((N,), torch.float32, 0.5, 3)
    out = eng.op_gemm(0, a, w, bias)
    ref = a.float() @ w.float().t() + bias
    assert within(rel_l2(out, ref), TOL[dtype_name])


def tile_rel_l2(out, ref, bm=256, bn=144):
    """Worst rel-L2 over the bm x bn output tiles: one wrong tile of a 8192 x 4608 product is 0.1 % of the elements and must not hide in the global norm."""
    M, N = ref.shape
    d = (out.float() - ref).pow(2)
    r = ref.pow(2)
    pm, pn = (-M) % bm, (-N) % bn
    d, r = F.pad(d, (0, pn, 0, pm)), F.pad(r, (0, pn, 0, pm))
    dn = d.view((M + pm) // bm, bm, (N + pn) // bn, bn).sum((1, 3))
    rn = r.view((M + pm) // bm, bm, (N + pn) // bn, bn).sum((1, 3))
    return float((dn / rn.clamp_min(1e-30)).sqrt().max())


# (kernel, call site, epilogue, M, N, K): the instantiations the engine launches, at the shapes it launches them with (config A: M = 8192) plus ragged M.
# wide288 = gemm2w.h (FF1 + GELU, attn1 K,V two-thirds of the fused QKV weight, hoisted text K/V of all 28 blocks); tile144 = gemm3s.h;
# tile128 = gemm.h (N % 144 != 0).
KERNEL_CASES = [
    ("wide288", "ff1", 1, 8192, 4608, 1152),
    ("wide288", "ff1", 1, 600, 4608, 1152),
    ("wide288", "attn1_kv", 0, 8192, 2304, 1152),
    ("wide288", "text_kv", 0, 3840, 64512, 1152),
    ("wide288", "text_kv", 0, 77, 2304, 1152),
    ("wide288", "small", 0, 1000, 576, 256),
    ("tile144", "qkv", 0, 8192, 1152, 1152),
    ("tile144", "qkv", 0, 8192, 3456, 1152),
    ("tile144", "attn2_q", 0, 700, 1152, 1152),
    ("tile144", "ff1", 1, 8192, 4608, 1152),
    ("tile144", "caption", 1, 3840, 1152, 4096),
    ("tile144", "small", 2, 40, 1152, 256),
    ("tile128", "small", 1, 1000, 1024, 512),
    ("tile144", "attn2_q", 0, 600, 1152, 192),  # three K-steps only: first / last-two steps of the staggered schedule with nothing in between
    # gemm3s.h latency variant: 64 x 144 tiles, four waves of 16 x 144 -- what the engine launches when a GEMM has few rows (batch 1-4)
    ("latency64", "attn2_q", 0, 512, 1152, 1152),
    ("latency64", "ff1", 1, 512, 4608, 1152),
    ("latency64", "attn1_kv", 0, 1000, 2304, 1152),
    ("latency64", "small", 2, 40, 1152, 256),
    ("latency64", "caption", 1, 240, 1152, 4096),
]


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("kernel,site,epi,M,N,K", KERNEL_CASES)
def test_gemm_every_kernel_by_call_site(dtype_name, kernel, site, epi, M, N, K):
    """Every GEMM kernel the engine can launch, through ecad_op_gemm_ex with the call site's own template instantiation; worst output TILE
    against fp32 torch (cached_transformer_block.py:348-353,374-382 are the Linear layers these replace)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    a, w = _r((M, K), dt, 1.0, 21), _r((N, K), dt, 0.05, 22)
    bias = _r((N,), torch.float32, 0.5, 23)
    out = eng.op_gemm_ex(kernel, site, epi, a, w, bias)
    ref = a.float() @ w.float().t() + bias
    ref = {0: lambda x: x, 1: lambda x: F.gelu(x, approximate="tanh"), 2: F.silu}[epi](ref)
    worst = tile_rel_l2(out, ref)
    del ref
    assert within(worst, TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("epi,M,N,K", [(1, 8192, 4608, 1152), (1, 600, 4608, 1152), (0, 1000, 2304, 1152), (0, 300, 576, 1152), (1, 65536, 576, 320), (1, 32768, 1152, 128)])
def test_gemm2w_direct_epilogue_stores_the_staged_epilogues_bytes(dtype_name, epi, M, N, K):
    """gemm2w.h (round 3): rows leave straight from the accumulator registers -- v_permlane16_swap between the packed registers of two neighbouring 16-column
    fragments turns a lane's two 8-byte pieces into one 16-byte piece, one store per fragment pair writes 64 contiguous bytes of 16 rows -- instead of through a
    C tile in LDS behind a barrier (the default for the GELU epilogue of FF1).  `sched` + 4 forces it for every epilogue, + 8 forces the staged path: the same
    bytes, ragged last tile and a strided output included, and nothing outside the output columns is touched.  With more tiles than CUs (a multiple of them) one
    workgroup per CU walks its tiles and requests the next tile's first K-step under the epilogue: 8192 x 4608 (2 tiles per workgroup), 65536 x 576 x 320 (2 tiles,
    an ODD number of K-steps: no early request, a barrier instead) and 32768 x 1152 x 128 (2 tiles of two K-steps)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    a, w = _r((M, K), dt, 1.0, 51), _r((N, K), dt, 0.05, 52)
    bias = _r((N,), torch.float32, 0.5, 53)
    outs = []
    for sched in (8, 4):
        buf = torch.full((M, N + 64), 3.0, dtype=dt, device="cuda")
        eng.op_gemm_ex("wide288", "ff1" if epi == 1 else ("attn1_kv" if N == 2304 else "small"), epi, a, w, bias, out=buf[:, 32 : 32 + N], sched=sched)
        assert bool((buf[:, :32] == 3.0).all()) and bool((buf[:, 32 + N :] == 3.0).all())
        outs.append(buf[:, 32 : 32 + N].clone())
    assert torch.equal(outs[0], outs[1])
    ref = a.float() @ w.float().t() + bias
    ref = F.gelu(ref, approximate="tanh") if epi == 1 else ref
    assert within(tile_rel_l2(outs[1], ref), TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("site,epi,M,N,K,sched", [("ff1", 1, 8192, 4608, 1152, 0), ("ff1", 1, 2048, 4608, 1152, 8), ("attn1_kv", 0, 8192, 2304, 1152, 0), ("attn1_kv", 0, 8192, 2304, 1152, 4),
                                                   ("small", 2, 512, 576, 256, 0), ("small", 0, 32768, 1152, 128, 4)])
def test_wide_tile_on_four_waves_stores_the_eight_wave_kernels_bytes(dtype_name, site, epi, M, N, K, sched, monkeypatch):
    """gemm2w4.h (round 5): the 256 x 288 tile on one wave per SIMD with the hand-placed K loop of scripts/gen_gemm4.py -- what the engine launches for whole 256-row
    tiles (ECAD_GEMM4=0: gemm2w.h's eight waves).  Same K order per accumulator and the same epilogue arithmetic: the same bytes, direct and staged epilogue, one tile
    per workgroup and two (8192 x 4608: 512 tiles on 256 CUs), a single trip of the loop (K = 256) and none (K = 128); both against fp32 torch."""
    from ecad_amd.engine import EcadEngine

    dt = getattr(torch, dtype_name)
    a, w = _r((M, K), dt, 1.0, 61), _r((N, K), dt, 0.05, 62)
    bias = _r((N,), torch.float32, 0.5, 63)
    outs = []
    for four in ("1", "0"):
        monkeypatch.setenv("ECAD_GEMM4", four)
        eng = EcadEngine(num_layers=1, max_batch=2, max_tokens=256, max_text_tokens=120, dtype=dt)
        buf = torch.full((M, N + 64), 3.0, dtype=dt, device="cuda")
        eng.op_gemm_ex("wide288", site, epi, a, w, bias, out=buf[:, 32 : 32 + N], sched=sched)
        assert bool((buf[:, :32] == 3.0).all()) and bool((buf[:, 32 + N :] == 3.0).all())
        outs.append(buf[:, 32 : 32 + N].clone())
        del eng
    assert torch.equal(outs[0], outs[1])
    ref = a.float() @ w.float().t() + bias
    ref = {0: lambda x: x, 1: lambda x: F.gelu(x, approximate="tanh"), 2: F.silu}[epi](ref)
    assert within(tile_rel_l2(outs[0], ref), TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("site,K,B,Ntok,gated", [("attn1_out", 1152, 32, 256, True), ("attn2_out", 1152, 8, 256, False), ("ff2", 4608, 32, 256, True), ("ff2", 4608, 2, 4096, True), ("attn1_out", 256, 4, 256, True),
                                                 ("attn2_out", 128, 4, 512, "table")])
def test_read_modify_write_epilogue_on_the_wide_tile_is_gemm3s_bit_for_bit(dtype_name, site, K, B, Ntok, gated):
    """gemm2w4.h's read-modify-write epilogue (round 5): cache <- out; hidden += gate * out (cached_transformer_block.py:243-246,317-320,355-360) on 256 x 288 tiles, the
    rounding of acc + bias into the C tile inside the generated statement.  What the engine launches for both out-projections and FF2 when the wide tiles come in
    whole rounds of the chip or cover half of it with another batch in flight -- so it must store gemm3s's bytes: cache slot and hidden states equal bit for bit
    (M = 8192: half-chip launch; 2 x 4096 tokens: the Sigma geometry; K = 256: a single trip of the loop), and both within tolerance of fp32 torch."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    D, M = 1152, B * Ntok
    a, w = _r((M, K), dt, 1.0, 71), _r((D, K), dt, 0.03, 72)
    bias = _r((D,), torch.float32, 0.1, 73)
    h0 = _r((M, D), dt, 1.0, 74)
    table = _r((D,), torch.float32, 1.0, 75) if gated else None
    mod = _r((B, 6 * D), dt, 1.0, 76) if gated is True else None   # "table": the gate table alone (no modulation rows), K = 128: no trip of the loop at all
    res = {}
    for kernel in ("wide288", "tile144"):
        h = h0.clone()
        cache = torch.zeros((M, D), dtype=dt, device="cuda")
        eng.op_gemm_ex(kernel, site, 3, a, w, bias, resid=h, gate_table=table, gate_mod=mod[:, 2 * D :] if mod is not None else None, gate_mod_stride=6 * D if mod is not None else 0, rows_per_batch=Ntok,
                       out=cache)
        res[kernel] = (cache, h)
    assert torch.equal(res["wide288"][0], res["tile144"][0]) and torch.equal(res["wide288"][1], res["tile144"][1])
    out_ref = (a.float() @ w.float().t() + bias).to(dt).float()
    gate = ((table[None] + mod[:, 2 * D : 3 * D].float()).repeat_interleave(Ntok, 0) if mod is not None else table[None]) if gated else 1.0
    assert within(tile_rel_l2(res["wide288"][0], out_ref), TOL[dtype_name])
    assert within(tile_rel_l2(res["wide288"][1], h0.float() + gate * out_ref), TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("site,K,B,Ntok,gated,has_bias", [("attn1_out", 1152, 32, 256, True, True), ("attn2_out", 1152, 8, 256, False, True), ("ff2", 4608, 32, 256, True, True),
                                                          ("ff2", 4608, 2, 4096, True, True), ("attn1_out", 192, 4, 256, True, False), ("attn2_out", 384, 4, 512, "table", True)])
def test_narrow_tile_on_four_waves_stores_gemm3s_bytes(dtype_name, site, K, B, Ntok, gated, has_bias):
    """gemm3s4.h (round 6): gemm3s's 256 x 144 tile with the read-modify-write epilogue -- cache <- out; hidden += gate * out (cached_transformer_block.py:243-246,
    317-320,355-360) -- on four waves, one per SIMD, the K loop a generated statement that also issues the residual loads and rounds acc + bias into the C tile.  The
    engine launches it for FF2 of a batch that runs alone, so it must store gemm3s's bytes: cache slot and hidden states bit for bit (K = 192: the single last trip of
    the loop; K = 384: one full trip + the last; no bias: the statement adds the zeros the kernel put there), and both within tolerance of fp32 torch.  The row
    statistics and replayed components are compared bit for bit by scripts/probe_gemm3s4.hip and, through whole plans, by the engine tests that run every
    read-modify-write launch on it (ECAD_GEMM3S4=3)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    D, M = 1152, B * Ntok
    a, w = _r((M, K), dt, 1.0, 81), _r((D, K), dt, 0.03, 82)
    bias = _r((D,), torch.float32, 0.1, 83) if has_bias else None
    h0 = _r((M, D), dt, 1.0, 84)
    table = _r((D,), torch.float32, 1.0, 85) if gated else None
    mod = _r((B, 6 * D), dt, 1.0, 86) if gated is True else None
    res = {}
    for kernel in ("narrow4", "tile144"):
        h = h0.clone()
        cache = torch.zeros((M, D), dtype=dt, device="cuda")
        eng.op_gemm_ex(kernel, site, 3, a, w, bias, resid=h, gate_table=table, gate_mod=mod[:, 2 * D :] if mod is not None else None, gate_mod_stride=6 * D if mod is not None else 0, rows_per_batch=Ntok,
                       out=cache)
        res[kernel] = (cache, h)
    assert torch.equal(res["narrow4"][0], res["tile144"][0]) and torch.equal(res["narrow4"][1], res["tile144"][1])
    out_ref = (a.float() @ w.float().t() + (bias if bias is not None else 0.0)).to(dt).float()
    gate = ((table[None] + mod[:, 2 * D : 3 * D].float()).repeat_interleave(Ntok, 0) if mod is not None else table[None]) if gated else 1.0
    assert within(tile_rel_l2(res["narrow4"][0], out_ref), TOL[dtype_name])
    assert within(tile_rel_l2(res["narrow4"][1], h0.float() + gate * out_ref), TOL[dtype_name])
    # it names its limits instead of substituting another kernel: whole 256-row tiles, K in whole trips of the three-stage loop, the three read-modify-write sites
    with pytest.raises((RuntimeError, ValueError), match="gemm3s4"):
        eng.op_gemm_ex("narrow4", site, 3, a[:, :128].contiguous(), w[:, :128].contiguous(), bias, resid=h0.clone(), rows_per_batch=Ntok, out=torch.zeros((M, D), dtype=dt, device="cuda"))
    with pytest.raises((RuntimeError, ValueError), match="gemm3s4"):
        eng.op_gemm_ex("narrow4", "ff1", 1, a, w, bias, out=torch.zeros((M, D), dtype=dt, device="cuda"))


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
def test_gemm_wide_kv_writes_into_fused_qkv_buffer(dtype_name):
    """The split QKV launch of the engine: Q on 256x144 tiles into columns [0, D), K,V on 256x288 tiles into columns [D, 3D) of one [M, 3D] buffer
    (row stride 3D on the output) -- together they must equal one fused projection, and neither may touch the other's columns."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    M, D = 8192, 1152
    a, w = _r((M, D), dt, 1.0, 31), _r((3 * D, D), dt, 0.05, 32)
    bias = _r((3 * D,), torch.float32, 0.5, 33)
    qkv = torch.full((M, 3 * D), 7.0, dtype=dt, device="cuda")
    eng.op_gemm_ex("wide288", "attn1_kv", 0, a, w[D:], bias[D:], out=qkv[:, D:])
    assert bool((qkv[:, :D] == 7.0).all())
    eng.op_gemm_ex("tile144", "qkv", 0, a, w[:D], bias[:D], out=qkv[:, :D])
    ref = a.float() @ w.float().t() + bias
    assert within(tile_rel_l2(qkv, ref), TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("kernel,site,K,B,Ntok", [("tile144", "attn1_out", 1152, 32, 256), ("tile144", "attn2_out", 1152, 3, 192), ("tile144", "ff2", 4608, 32, 256),
                                                 ("tile128", "small", 1152, 3, 192),
                                                 ("latency64", "attn1_out", 1152, 2, 256), ("latency64", "attn2_out", 1152, 3, 192), ("latency64", "ff2", 4608, 4, 256)])
def test_gemm_cache_gate_residual_every_kernel(dtype_name, kernel, site, K, B, Ntok):
    """cache <- out; hidden += gate * out with gates of O(1) (cached_transformer_block.py:243-246,317-320,355-360) on every kernel that carries that epilogue, at
    config A's shape (M = 8192: one sample per 256-row tile -> gate via LDS) and with 192 tokens per sample (tiles straddle samples -> per-chunk gate)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    D = 1152 if kernel != "tile128" else 1024
    M = B * Ntok
    a, w = _r((M, K), dt, 1.0, 41), _r((D, K), dt, 0.03, 42)
    bias = _r((D,), torch.float32, 0.1, 43)
    h0 = _r((M, D), dt, 1.0, 44)
    table = _r((D,), torch.float32, 1.0, 45)
    mod = _r((B, 6 * D), dt, 1.0, 46)
    h = h0.clone()
    cache = torch.zeros((M, D), dtype=dt, device="cuda")
    eng.op_gemm_ex(kernel, site, 3, a, w, bias, resid=h, gate_table=table, gate_mod=mod[:, 2 * D :], gate_mod_stride=6 * D, rows_per_batch=Ntok, out=cache)
    out_ref = (a.float() @ w.float().t() + bias).to(dt).float()
    gate = (table[None] + mod[:, 2 * D : 3 * D].float()).repeat_interleave(Ntok, 0)
    h_ref = h0.float() + gate * out_ref
    assert within(tile_rel_l2(cache, out_ref), TOL[dtype_name])
    assert within(tile_rel_l2(h, h_ref), TOL[dtype_name])


def test_gemm_ex_refuses_instead_of_substituting():
    eng = small_engine("float16")
    a, w = _r((256, 1152), torch.float16, 1.0, 1), _r((1152, 1152), torch.float16, 0.05, 2)
    with pytest.raises(ValueError, match="288"):
        eng.op_gemm_ex("wide288", "ff1", 1, a, w[:1008], None)  # N = 1008 = 7 * 144: not a multiple of 288
    with pytest.raises(ValueError, match="call sites"):
        eng.op_gemm_ex("wide288", "attn1_out", 0, a, w, None)
    with pytest.raises(ValueError, match="144"):
        eng.op_gemm_ex("tile144", "qkv", 0, a, w[:1024], None)
    with pytest.raises(ValueError, match="retired"):  # pgemm.h left the library in round 3: its id is refused, not mapped to another kernel
        eng.GEMM_KERNELS["persistent"] = 1
        try:
            eng.op_gemm_ex("persistent", "ff2", 0, a, w, None)
        finally:
            del eng.GEMM_KERNELS["persistent"]
    with pytest.raises(ValueError, match="attention tail"):
        eng.op_gemm_ex("latency64", "qkv", 4, a, w, None)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("site,epi,M,K", [("attn2_q", 0, 1024, 1152), ("attn1_out", 3, 512, 1152), ("ff2", 3, 768, 4608), ("caption", 1, 480, 4096)])
def test_every_tile_shape_of_a_call_site_sums_k_in_the_same_order(dtype_name, site, epi, M, K):
    """The 256-row tile and the 64-row latency tile of a call site must give the SAME bits (one accumulator per output, K in order, the same epilogue arithmetic), so
    that a row's value does not depend on how many rows travel with it (few-rows rule, tail split, batch 1 vs batch 16)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    D = 1152
    a, w = _r((M, K), dt, 1.0, 61), _r((D, K), dt, 0.04, 62)
    bias = _r((D,), torch.float32, 0.3, 63)
    outs = {}
    for kernel in ("tile144", "latency64"):
        if epi == 3:
            h = _r((M, D), dt, 1.0, 64)
            table = _r((D,), torch.float32, 1.0, 65)
            cache = torch.zeros((M, D), dtype=dt, device="cuda")
            eng.op_gemm_ex(kernel, site, 3, a, w, bias, resid=h, gate_table=table, rows_per_batch=256, out=cache)
            outs[kernel] = (cache.clone(), h.clone())
        else:
            outs[kernel] = (eng.op_gemm_ex(kernel, site, epi, a, w, bias).clone(),)
    for x, y in zip(outs["tile144"], outs["latency64"]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("M,gated", [(512, True), (192, False), (1024, True), (1536, False)])
def test_split_k_latency_tiles_against_fp32_and_the_sequential_kernel(dtype_name, M, gated):
    """Kernel id `latency64_splitk` (what the engine launches for FF2 with few rows): four K-slices per 64 x 144 tile into fp32 partial sums, then splitk_rmw_kernel.
    Cache slot and updated hidden states per output tile against fp32, and against the sequential-K latency tile within one output rounding."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    D, K = 1152, 4608
    a, w = _r((M, K), dt, 1.0, 71), _r((D, K), dt, 0.03, 72)
    bias = _r((D,), torch.float32, 0.3, 73)
    h0 = _r((M, D), dt, 1.0, 74)
    table = _r((D,), torch.float32, 1.0, 75) if gated else None
    outs = {}
    for kernel in ("latency64_splitk", "latency64"):
        h = h0.clone()
        cache = torch.zeros((M, D), dtype=dt, device="cuda")
        eng.op_gemm_ex(kernel, "ff2", 3, a, w, bias, resid=h, gate_table=table, rows_per_batch=64, out=cache)
        outs[kernel] = (cache.clone(), h.clone())
    out_ref = a.float() @ w.float().t() + bias
    h_ref = h0.float() + (table if gated else 1.0) * out_ref.to(dt).float()
    c, h = outs["latency64_splitk"]
    assert within(tile_rel_l2(c, out_ref), TOL[dtype_name])
    assert within(tile_rel_l2(h, h_ref), TOL[dtype_name])
    assert not torch.equal(c, outs["latency64"][0])  # another summation order: this really is the split kernel
    assert within(tile_rel_l2(c, outs["latency64"][0].float()), {"float16": 4.5e-5, "bfloat16": 2.4e-4}[dtype_name])  # measured 2.3e-5 / 1.2e-4: a fraction of one output rounding
    with pytest.raises(ValueError, match="read-modify-write"):
        eng.op_gemm_ex("latency64_splitk", "ff2", 0, a, w, bias)


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric W catches a transposed C write (guide 3, 'A=I check')."""
    eng = small_engine("float16")
    K = N = 128
    a = torch.eye(128, dtype=torch.float16).cuda()
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 / 251.0).half().cuda()
    out = eng.op_gemm(0, a, w, None)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), w.t().contiguous().cpu())


@pytest.mark.parametrize("epi,fn", [(1, lambda x: F.gelu(x, approximate="tanh")), (2, F.silu)])
def test_gemm_activation(epi, fn):
    eng = small_engine("float16")
    a, w = _r((256, 1152), torch.float16, 1.0, 4), _r((1152, 1152), torch.float16, 0.06, 5)
    bias = _r((1152,), torch.float32, 0.5, 6)
    out = eng.op_gemm(epi, a, w, bias)
    ref = fn(a.float() @ w.float().t() + bias)
    assert within(rel_l2(out, ref), TOL["float16"])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("gated", [True, False])
def test_gemm_cache_gate_residual(dtype_name, gated):
    """cache <- out; hidden += gate * out  (cached_transformer_block.py:243-246,355-360)."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    B, Ntok, D, K = 3, 256, 1152, 1152
    M = B * Ntok
    a, w = _r((M, K), dt, 1.0, 7), _r((D, K), dt, 0.03, 8)
    bias = _r((D,), torch.float32, 0.1, 9)
    h0 = _r((M, D), dt, 1.0, 10)
    table = _r((D,), torch.float32, 0.3, 11) if gated else None
    mod = _r((B, 6 * D), dt, 0.3, 12) if gated else None
    h = h0.clone()
    cache = torch.zeros((M, D), dtype=dt, device="cuda")
    eng.op_gemm(3, a, w, bias, resid=h, gate_table=table, gate_mod=(mod[:, 2 * D :] if gated else None), gate_mod_stride=6 * D, rows_per_batch=Ntok, out=cache)
    out_ref = (a.float() @ w.float().t() + bias).to(dt).float()
    gate = (table[None] + mod[:, 2 * D : 3 * D].float()).repeat_interleave(Ntok, 0) if gated else 1.0
    h_ref = h0.float() + gate * out_ref
    assert within(rel_l2(cache, out_ref), TOL[dtype_name])
    assert within(rel_l2(h, h_ref), TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
def test_ln_modulate(dtype_name):
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    B, Ntok, D = 3, 100, 1152
    x = _r((B * Ntok, D), dt, 2.0, 1) + 0.5
    tsh, tsc = _r((D,), torch.float32, 0.3, 2), _r((D,), torch.float32, 0.3, 3)
    mod = _r((B, 6 * D), dt, 0.3, 4)
    y = eng.op_ln_modulate(x, tsh, tsc, mod[:, 0:], mod[:, D:], 6 * D, Ntok)
    n = F.layer_norm(x.float(), (D,), eps=1e-6)
    sh = (tsh[None] + mod[:, :D].float()).repeat_interleave(Ntok, 0)
    sc = (tsc[None] + mod[:, D : 2 * D].float()).repeat_interleave(Ntok, 0)
    assert within(rel_l2(y, n * (1 + sc) + sh), TOL[dtype_name])


@pytest.mark.parametrize("n_src,B,Ntok", [(1, 2, 256), (3, 2, 256), (9, 2, 256), (84, 2, 256), (5, 3, 6), (7, 1, 1022)])
def test_cache_accumulate(n_src, B, Ntok):
    """h += sum gate_i * C_i; gate-less sources mimic attn2 (cached_transformer_block.py:289).  Token counts divisible by 4 take the
    4-rows-per-thread kernel, ragged ones (6, 1022 tokens per sample) the one-chunk-per-thread kernel."""
    eng = small_engine("float16")
    D = 1152
    M = B * Ntok
    h0 = _r((M, D), torch.float16, 1.0, 1)
    srcs = [_r((M, D), torch.float16, 0.2, 10 + i) for i in range(n_src)]
    mod = _r((B, 6 * D), torch.float16, 0.3, 5)
    tables = [None if i % 3 == 1 else _r((D,), torch.float32, 0.3, 100 + i) for i in range(n_src)]
    mods = [None if i % 3 == 1 else (mod[:, 2 * D :] if i % 3 == 0 else mod[:, 5 * D :]) for i in range(n_src)]
    h = h0.clone()
    eng.op_cache_accumulate(h, srcs, tables, mods, 6 * D, Ntok)
    ref = h0.float()
    for i in range(n_src):
        if tables[i] is None:
            ref = ref + srcs[i].float()
        else:
            g = (tables[i][None] + mods[i][:, :D].float()).repeat_interleave(Ntok, 0)
            ref = ref + g * srcs[i].float()
    assert within(rel_l2(h, ref), TOL["float16"])


def _attn_ref(q, k, v, B, H, Nq, Nk, dh, bias):
    qh = q.float().view(B, Nq, H, dh).transpose(1, 2)
    kh = k.float().view(B, Nk, H, dh).transpose(1, 2)
    vh = v.float().view(B, Nk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if bias is not None:
        s = s + bias[:, None, None, :]
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Nq, H * dh)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("Nq,Nk", [(256, 256), (64, 64), (1024, 1024), (100, 100)])
def test_self_attention(dtype_name, Nq, Nk):
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    B, H, dh = 2, 16, 72
    qkv = _r((B * Nq, 3 * H * dh), dt, 1.0, 1)  # fused [M, 3D] buffer as the engine lays it out
    D = H * dh
    out = eng.op_attention(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, Nq, Nk, dh)
    ref = _attn_ref(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, Nq, Nk, dh, None)
    assert within(rel_l2(out, ref), ATT_TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("dh,Nq,Nk,masked", [(72, 256, 256, False), (72, 256, 120, True), (64, 256, 256, False), (64, 200, 77, True), (40, 1024, 1024, False), (8, 64, 64, False)])
def test_attention_generic_kernel(dtype_name, dh, Nq, Nk, masked):
    """attention.h (two-rounds kernel): what the engine launches whenever head_dim != 72 and under ECAD_ATTN=generic; forced here through ecad_op_attention_ex."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    B, H = 2, 4
    D = H * dh
    q = _r((B * Nq, D), dt, 1.0, 1)
    kv = _r((B * Nk, 2 * D), dt, 1.0, 2)
    bias = None
    if masked:
        lens = torch.tensor([Nk, max(1, Nk // 3)])
        bias = ((1 - (torch.arange(Nk)[None] < lens[:, None]).float()) * -10000.0).cuda()
    out = eng.op_attention(q, kv[:, :D], kv[:, D:], B, H, Nq, Nk, dh, bias=bias, kernel="generic")
    ref = _attn_ref(q, kv[:, :D], kv[:, D:], B, H, Nq, Nk, dh, bias)
    assert within(rel_l2(out, ref), ATT_TOL[dtype_name])
    if dh != 72:
        with pytest.raises(ValueError, match="head_dim 72"):
            eng.op_attention(q, kv[:, :D], kv[:, D:], B, H, Nq, Nk, dh, bias=bias, kernel="shared")
        auto = eng.op_attention(q, kv[:, :D], kv[:, D:], B, H, Nq, Nk, dh, bias=bias)  # the engine's own choice for this head_dim is this kernel
        assert torch.equal(auto, out)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("kind,Bt,Ntok,Nk", [("self", 3, 256, 256), ("self", 2, 1024, 1024), ("self", 1, 4096, 4096), ("cross", 3, 256, 120), ("cross", 2, 1024, 300),
                                            ("cross", 2, 256, 1)])
def test_query_projection_with_attention_in_its_tail(monkeypatch, dtype_name, kind, Bt, Ntok, Nk):
    """gemm3s_attn.h through ecad_op_qproj_attention: the launch the engine uses for attention on the 256 / 1024 / 4096-token geometries -- the Q projection
    (256 x 144 tiles = one sample x two heads) with softmax(Q K^T / sqrt(72) + bias) V of its rows in the GEMM tail.  Reference: fp32 projection of the
    rounded inputs, Q rounded to the storage type (the value the separate launches pass through HBM), fp32 attention.  Judged per 256 x 144 output tile
    (= per workgroup), worst tile.  Self: K / V are column slices of a fused [M, 3D] buffer as the engine lays it out; cross: a [Bt*T, 2D] slice of a wider
    buffer, ragged key counts, additive -10000 mask, one sample with a single live key."""
    eng = small_engine(dtype_name)
    dt = getattr(torch, dtype_name)
    H, dh = 16, 72
    D = H * dh
    x = _r((Bt * Ntok, D), dt, 1.0, 1)
    wq = _r((D, D), dt, 0.03, 2)
    bq = (_r((D,), torch.float32, 0.1, 3)).float()
    if kind == "self":
        buf = _r((Bt * Ntok, 3 * D), dt, 1.0, 4)
        k, v, bias, site = buf[:, D : 2 * D], buf[:, 2 * D :], None, "qkv"
    else:
        buf = _r((Bt * Nk, 4 * D), dt, 1.0, 4)  # wider than [K | V]: the row stride is not 2 D
        k, v, site = buf[:, D : 2 * D], buf[:, 2 * D : 3 * D], "attn2_q"
        lens = torch.tensor([Nk, max(1, Nk // 3), 1][:Bt])
        bias = ((1 - (torch.arange(Nk)[None] < lens[:, None]).float()) * -10000.0).cuda()
    out = eng.op_qproj_attention(site, x, wq, bq, k, v, Bt, Ntok, Nk, key_bias=bias)
    q = (x.float() @ wq.float().t() + bq).to(dt)
    ref = _attn_ref(q, k, v, Bt, H, Ntok, Nk, dh, bias)
    assert torch.isfinite(out).all()
    assert within(tile_rel_l2(out, ref), ATT_TOL[dtype_name])
    # and it is the same arithmetic as projection + attention as two launches
    q2 = eng.op_gemm_ex("tile144", site, 0, x, wq, bq)
    two = eng.op_attention(q2, k, v, Bt, H, Ntok, Nk, dh, bias=bias, kernel="shared")
    assert torch.equal(out, two)
    # few rows (all of these cases): the product default runs the same tail on 64 x 144 latency tiles -- 64 query rows x two heads per workgroup
    monkeypatch.setenv("ECAD_LATENCY_TILES", "1")
    lat = small_engine(dtype_name)
    assert torch.equal(lat.op_qproj_attention(site, x, wq, bq, k, v, Bt, Ntok, Nk, key_bias=bias), out)


def test_query_projection_with_attention_refuses_other_geometries():
    eng = small_engine("float16")
    D = 1152
    x = _r((240, D), torch.float16, 1.0, 1)
    wq = _r((D, D), torch.float16, 0.03, 2)
    kv = _r((240, 2 * D), torch.float16, 1.0, 3)
    with pytest.raises(ValueError, match="n_tokens % 256"):
        eng.op_qproj_attention("qkv", x, wq, None, kv[:, :D], kv[:, D:], 1, 240, 240)
    with pytest.raises(ValueError, match="query projections"):
        eng.op_qproj_attention("ff1", x, wq, None, kv[:, :D], kv[:, D:], 1, 240, 240)


@pytest.mark.parametrize("T", [120, 300, 1])
def test_cross_attention_masked(T):
    """-10000 additive bias on discarded text tokens (pixart_transformer_2d_edited.py:282-289)."""
    eng = small_engine("float16")
    B, H, dh, Nq = 3, 16, 72, 256
    D = H * dh
    q = _r((B * Nq, D), torch.float16, 1.0, 1)
    kv = _r((B * T, 2 * D), torch.float16, 1.0, 2)
    lens = torch.tensor([T, max(1, T // 3), 1])
    mask = (torch.arange(T)[None] < lens[:, None]).float()
    bias = ((1 - mask) * -10000.0).cuda()
    out = eng.op_attention(q, kv[:, :D], kv[:, D:], B, H, Nq, T, dh, bias=bias)
    ref = _attn_ref(q, kv[:, :D], kv[:, D:], B, H, Nq, T, dh, bias)
    assert within(rel_l2(out, ref), ATT_TOL["float16"])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("N", [1024, 2048])
def test_self_attention_four_fragment_kernel_is_the_same_arithmetic(dtype_name, N):
    """attention_dma.h with four query fragments per wave (512 query rows per work item; K / V chunks by LDS-DMA into a double-buffered image whose pads -- zeros
    and the ones column -- are written once and never touched again): what the plans launch for self-attention over >= 1024 tokens (Sigma-1024, alpha-512)
    instead of the fused tail.  Same fragment layouts and operation order as attention128.h: bit-identical outputs, with a late dominant key tile (running-
    maximum rescale), and both within the attention tolerance of an fp32 reference."""
    dt = getattr(torch, dtype_name)
    B, H, dh = 3, 16, 72
    D = H * dh
    qkv = _r((B * N, 3 * D), dt, 1.0, 5)
    qkv[N - 40 : N - 32, D : 2 * D] *= 5.0
    eng = small_engine(dtype_name)
    want = eng.op_attention(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, N, N, dh, kernel="shared")
    got = eng.op_attention(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, N, N, dh, kernel="wide4")
    assert torch.equal(got, want)
    assert within(rel_l2(got, _attn_ref(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, N, N, dh, None)), ATT_TOL[dtype_name])


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("N", [256, 1024, 4096])
def test_self_attention_on_one_wave_per_simd_for_heads_of_72(dtype_name, N):
    """attention_w472.h (round 6): d_h 72 on the one-wave-per-SIMD structure of the FLUX kernel -- a generated instruction stream (scripts/gen_attn_w4_72.py), five
    k-steps of S = K Q~^T with the last half zero, the row sums as row 72 of O^T (a ones column preset in the V ring that the masked DMA never overwrites), deferred
    rescale with threshold 2^8.  What the plans launch for self-attention over >= 1024 tokens (Sigma-1024: N = 4096; 634 against 813 us for attention_dma.h's
    four-fragment kernel).  Another MFMA shape and summation order than attention128.h: not bit-identical, held to the fp32 reference at the attention tolerance --
    with a late dominant key tile (the recentre / rescale paths), large scores, and launch-to-launch reproducibility (a race between the DMA ring and the fragment
    reads shows up there first)."""
    dt = getattr(torch, dtype_name)
    B, H, dh = (2 if N == 4096 else 3), 16, 72
    D = H * dh
    eng = small_engine(dtype_name)
    for case in ("plain", "late dominant keys", "large scores"):
        qkv = _r((B * N, 3 * D), dt, 1.0, 7)
        if case == "late dominant keys":
            qkv[N - 40 : N - 32, D : 2 * D] *= 5.0
        if case == "large scores":
            qkv[:, : 2 * D] *= 1.7
        q, k, v = qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :]
        got = eng.op_attention(q, k, v, B, H, N, N, dh, kernel="w472")
        again = eng.op_attention(q, k, v, B, H, N, N, dh, kernel="w472")
        assert torch.equal(got, again)
        assert torch.isfinite(got.float()).all()
        assert within(rel_l2(got, _attn_ref(q, k, v, B, H, N, N, dh, None)), ATT_TOL[dtype_name]), case
    with pytest.raises(Exception):
        eng.op_attention(qkv[: B * 128, :D], qkv[: B * 128, D : 2 * D], qkv[: B * 128, 2 * D :], B, H, 128, 128, dh, kernel="w472")   # Nq % 256 != 0: refused, not mis-run


def test_attention_large_scores_online_softmax():
    """Force the running-max rescale branch: one key tile dominates late (guide rule 26)."""
    eng = small_engine("float16")
    B, H, dh, N = 1, 16, 72, 256
    D = H * dh
    qkv = _r((N, 3 * D), torch.float16, 1.0, 3)
    qkv[200:208, D : 2 * D] *= 6.0  # keys in the 4th tile get much larger logits
    out = eng.op_attention(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, N, N, dh)
    ref = _attn_ref(qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :], B, H, N, N, dh, None)
    assert within(rel_l2(out, ref), ATT_TOL["float16"])
