"""GPU parity tests of every C-ABI kernel against the oracle (float64 CPU) — run on the MI355X box with -m gpu."""
import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from rat_amd._lib import get_lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return get_lib()


@pytest.mark.parametrize("d", [8, 10, 64])
def test_gather_fwd_bwd(lib, d):
    kc.check_gather(lib, "cuda", d)
    kc.check_gather(lib, "cuda", d, B=37, T=11)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("shape", [(70, 37, 29), (512, 400, 1280), (4096, 1, 400), (400, 1280, 512), (400, 400, 4096), (1, 400, 4096)])
def test_sgemm(lib, ta, tb, shape):
    kc.check_sgemm(lib, "cuda", ta, tb, *shape)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("shape", [(72, 40, 100), (512, 400, 1280), (400, 1280, 512), (400, 400, 4096), (4096, 400, 400), (100, 240, 64)])
def test_sgemm_bf16x3(lib, ta, tb, shape):
    kc.check_sgemm(lib, "cuda", ta, tb, *shape, arith="bf16x3")


ATTN_CASES = [  # B, T, S, d, heads, dh, project_out
    (2, 3, 4, 8, 2, 4, True),
    (1, 4, 5, 10, 2, 10, True),
    (2, 2, 3, 8, 1, 8, False),
    (3, 11, 21, 64, 8, 10, True),       # north-star shape
    (5, 6, 14, 40, 8, 10, True),        # KKBox config shape
    (4, 31, 9, 10, 4, 10, True),        # Tmall-like: long cross sequences
    (300, 6, 4, 10, 2, 10, True),       # ML-Tag config shape, many chunks per work-group
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=str)
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_fwd_bwd(lib, case, mode):
    kc.check_attn(lib, "cuda", case, mode)


# bf16x3 arithmetic (north-star geometry only): one chunk, many chunks per work-group with a ragged last one, both sequence
# lengths of the north star, the KKBox-like S = 14 and Tmall-like T = 31 lengths — same tolerances as the exact-fp32 kernels, plus
# agreement of the two arithmetic variants to 4e-6 of each tensor's largest element
B3_CASES = [(3, 11, 21, 64, 8, 10, True), (300, 11, 21, 64, 8, 10, True), (37, 11, 14, 64, 8, 10, True), (29, 31, 9, 64, 8, 10, True),
            (1, 2, 64, 64, 8, 10, True),
            # one sequence per chunk (33 ... 48 tokens; BASELINE configs[3]: F = 40 -> 41): one chunk, many chunks per work-group with a ragged
            # last round, both phases; forward core on the matrix pipe from 40 tokens on (three 16-row tiles), backward on the VALU
            (1, 33, 41, 64, 8, 10, True), (130, 11, 41, 64, 8, 10, True), (7, 48, 33, 64, 8, 10, True)]


@pytest.mark.parametrize("case", B3_CASES, ids=str)
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_fwd_bwd_bf16x3(lib, case, mode):
    kc.check_attn(lib, "cuda", case, mode, arith="bf16x3")


@pytest.mark.parametrize("d", [40, 48, 56])
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_fwd_bwd_bf16x3_narrower_embedding(lib, d, mode):
    """embedding_dim 40 (the shipped KKBox config) / 48 / 56 inside the 64-wide tiles of the bf16x3 kernels; many chunks per work-group"""
    kc.check_attn(lib, "cuda", (40, 6, 14, d, 8, 10, True), mode, arith="bf16x3")
    kc.check_attn(lib, "cuda", (3, 5, 3, d, 8, 10, True), mode, arith="bf16x3")


def test_attn_bwd_bf16x3_probabilities_handed_to_pass_two(lib, knob):
    """for L <= 12 pass 1 of attn_bwd3_kernel leaves P in LDS and pass 2 reads it (the default; knob attn_bwd_ph = 0 recomputes) — both forms"""
    kc.check_attn(lib, "cuda", (40, 11, 21, 64, 8, 10, True), "cross", arith="bf16x3")
    kc.check_attn(lib, "cuda", (7, 3, 12, 64, 8, 10, True), "intra", arith="bf16x3")
    knob(lib, "attn_bwd_ph", 0)
    kc.check_attn(lib, "cuda", (40, 11, 21, 64, 8, 10, True), "cross", arith="bf16x3")


@pytest.mark.parametrize("case,mode", [((40, 6, 21, 64, 8, 10, True), "intra"), ((30, 11, 4, 64, 8, 10, True), "cross"), ((9, 31, 9, 64, 8, 10, True), "cross"),
                                       ((9, 31, 16, 64, 8, 10, True), "intra"), ((5, 28, 3, 64, 8, 10, True), "cross"), ((7, 3, 32, 64, 8, 10, True), "intra")],
                         ids=["L21", "L11", "L31", "L16", "L28", "L32"])
def test_attn_bwd_bf16x3_matrix_pipe_core(lib, case, mode, knob):
    """the backward core on the matrix pipe (attn_bwd3_kernel<.., MC>): forced on at every length class (ragged tiles, full tiles, one
    and two tiles per sequence) and — L 31, 28, 32 — selected by the host's own rule"""
    knob(lib, "attn_bwd_core_mfma", 1)
    kc.check_attn(lib, "cuda", case, mode, arith="bf16x3")
    knob(lib, "attn_bwd_core_mfma", -1)
    kc.check_attn(lib, "cuda", case, mode, arith="bf16x3")


@pytest.mark.parametrize("case,mode", [((60, 11, 41, 64, 8, 10, True), "intra"), ((5, 48, 3, 64, 8, 10, True), "cross"), ((3, 2, 33, 64, 8, 10, True), "intra")],
                         ids=["L41", "L48", "L33"])
def test_attn_bwd_bf16x3_matrix_pipe_core_three_tiles(lib, case, mode, knob):
    """33 ... 48 tokens (one sequence per chunk, three 16-row tiles): the key-tile-inner form of the matrix core (b3_bwd_core_mfma_kt), forced by
    the knob (the host's own rule takes it from 40 tokens on: test_attn_fwd_bwd_bf16x3's 41-token cases run it by default)"""
    knob(lib, "attn_bwd_core_mfma", 1)
    kc.check_attn(lib, "cuda", case, mode, arith="bf16x3")
    kc.check_attn_dropout(lib, "cuda", case, mode, arith="bf16x3")


@pytest.mark.parametrize("case,mode", [((40, 6, 21, 64, 8, 10, True), "intra"), ((30, 11, 4, 64, 8, 10, True), "cross"), ((9, 31, 9, 64, 8, 10, True), "cross"),
                                       ((9, 31, 16, 64, 8, 10, True), "intra"), ((5, 28, 3, 64, 8, 10, True), "cross"), ((7, 3, 32, 64, 8, 10, True), "intra"),
                                       ((60, 11, 41, 64, 8, 10, True), "intra"), ((5, 48, 3, 64, 8, 10, True), "cross"), ((4, 40, 2, 64, 8, 10, True), "cross"),
                                       ((3, 2, 33, 64, 8, 10, True), "intra")],
                         ids=["L21", "L11", "L31", "L16", "L28", "L32", "L41_three_tiles", "L48", "L40", "L33_forced"])
def test_attn_fwd_exact_fp32_matrix_pipe_core(lib, case, mode, knob):
    """the forward core on the matrix pipe (attn_fwd3_kernel<.., MCF>): forced on at every length class, forced off, and by the host's own
    rule (L 28 ... 32); also through the entry point with a residual of its own (the EX instantiation) and with dropout"""
    for k in (2, 3, 0):
        knob(lib, "attn_fwd_core_mfma", k)
        kc.check_attn(lib, "cuda", case, mode, arith="bf16x3")
    knob(lib, "attn_fwd_core_mfma", 2)
    kc.check_attn_dropout(lib, "cuda", case, mode, arith="bf16x3")


def test_attn_narrower_embedding_with_queries_and_dropout(lib):
    kc.check_attn_queries(lib, "cuda", (30, 6, 14, 40, 8, 10, True), "intra", nq=1, arith="bf16x3")
    kc.check_attn_queries(lib, "cuda", (30, 6, 14, 40, 8, 10, True), "cross", nq=1, arith="bf16x3")
    kc.check_attn_dropout(lib, "cuda", (4, 6, 14, 40, 8, 10, True), "intra", arith="bf16x3")
    kc.check_attn_dropout(lib, "cuda", (4, 6, 14, 40, 8, 10, True), "cross", arith="bf16x3")


@pytest.mark.parametrize("case,mode,nq,arith", [((4, 11, 21, 64, 8, 10, True), "intra", 1, "bf16x3"), ((40, 11, 21, 64, 8, 10, True), "intra", 1, "bf16x3"),
                                                ((30, 11, 4, 64, 8, 10, True), "cross", 1, "bf16x3"), ((2, 3, 7, 64, 8, 10, True), "intra", 3, "bf16x3"),
                                                ((2, 3, 4, 8, 2, 4, True), "intra", 1, "f32"), ((2, 4, 5, 64, 8, 10, True), "cross", 1, "f32")],
                         ids=["b3_intra_L21", "b3_intra_many_chunks", "b3_cross_L11", "b3_three_queries", "generic_ignores", "fast_f32_ignores"])
def test_attn_with_a_subset_of_query_positions(lib, case, mode, nq, arith):
    """RatSeqMap.queries (the last encoder block's dead-token pruning): the bf16x3 kernels skip the other queries"""
    kc.check_attn_queries(lib, "cuda", case, mode, nq=nq, arith=arith)


@pytest.mark.parametrize("nseq,L,heads,dh,softmax_scale", [(2, 5, 2, 4, None), (7, 231, 8, 10, None), (64, 84, 8, 10, None), (3, 400, 2, 20, 0.2), (5, 33, 2, 7, 0.3), (2, 600, 1, 16, None),
                                                           (5, 48, 3, 10, 0.3), (3, 900, 2, 10, None), (300, 60, 32, 10, None),
                                                           # round 6 (the matrix-pipe backward): 7 key tiles = odd against 3 per trip / 2 per group, a
                                                           # ragged last tile; 32 full tiles; the longest sequence whose four row tiles fit the LDS
                                                           (3, 97, 2, 10, None), (4, 512, 3, 10, None), (2, 799, 1, 10, 0.2)])
def test_attn_core_fwd_bwd(lib, nseq, L, heads, dh, softmax_scale, knob):
    """(dim_head 10 with 48 ... 1024 tokens: the forward runs on the matrix pipe — core_fwd_mfma_kernel; the knob's value 3 keeps the VALU kernel;
    up to ~800 tokens the backward is the hybrid kernel — core_bwd_hybrid_kernel: dQ on the matrix pipe beside dK / dV on the VALU — and
    attn_bwd_core_mfma = 0 keeps the two VALU passes: every case with the knobs both ways)"""
    kc.check_attn_core(lib, "cuda", nseq, L, heads, dh, softmax_scale)
    if dh == 10 and L >= 48:
        knob(lib, "attn_fwd_core_mfma", 3)
        knob(lib, "attn_bwd_core_mfma", 0)
        kc.check_attn_core(lib, "cuda", nseq, L, heads, dh, softmax_scale)


@pytest.mark.parametrize("B,T,S,heads,dh", [(2, 3, 4, 2, 4), (64, 31, 9, 32, 10), (16, 11, 21, 8, 10), (6, 61, 5, 8, 10), (40, 231, 3, 8, 10)])
def test_attn_core_strided(lib, B, T, S, heads, dh, knob):
    """(T = 61 / 231 at dim_head 10: the matrix-pipe forward and the hybrid backward through a strided RatSeqMap; more pairs than work-groups
    with max_blocks = 1)"""
    kc.check_attn_core_strided(lib, "cuda", B, T, S, heads, dh)
    if T >= 48:
        knob(lib, "max_blocks", 1)
        kc.check_attn_core_strided(lib, "cuda", B, T, S, heads, dh)


@pytest.mark.parametrize("case,mode,res_mode,dropout", [pytest.param((40, 11, 21, 64, 16, 10, True), "intra", "x", 0.0, id="G2_intra_L21"),
                                                        pytest.param((40, 11, 21, 64, 32, 10, True), "cross", "x", 0.0, id="G4_cross_L11"),
                                                        pytest.param((30, 31, 9, 64, 32, 10, True), "cross", "other", 0.25, id="G4_cross_L31_dropout"),
                                                        pytest.param((30, 31, 9, 64, 32, 10, True), "intra", "x", 0.1, id="G4_intra_L9_dropout"),
                                                        pytest.param((7, 3, 64, 64, 64, 10, True), "intra", "x", 0.0, id="G8_L64")])
def test_attn_wide_heads_group_loop(lib, case, mode, res_mode, dropout):
    """rat_attn_fwd_groups (attn_fwd3_kernel<GRP>): every head group of a chunk inside one launch"""
    kc.check_attn_groups(lib, "cuda", case, mode, res_mode=res_mode, dropout=dropout)


@pytest.mark.parametrize("case,mode,res_mode,dropout", [pytest.param((300, 6, 10, 10, 32, 10, True), "intra", "x", 0.0, id="tmall_G4_intra_L10"),
                                                        pytest.param((300, 6, 10, 10, 32, 10, True), "cross", "x", 0.0, id="tmall_G4_cross_L6"),
                                                        pytest.param((40, 13, 5, 10, 16, 10, True), "cross", "other", 0.25, id="G2_cross_L13_dropout"),
                                                        pytest.param((40, 5, 21, 16, 24, 10, True), "intra", "other", 0.1, id="G3_d16_L21_dropout"),
                                                        pytest.param((9, 3, 64, 12, 64, 10, True), "intra", "x", 0.0, id="G8_forward_only"),
                                                        # round 6: groups of 4 heads x 20 — RAT_m3 at the Tmall geometry (16 heads of width 20)
                                                        pytest.param((300, 6, 10, 10, 16, 20, True), "intra", "x", 0.0, id="m3_tmall_G4_intra_L10"),
                                                        pytest.param((300, 6, 10, 10, 16, 20, True), "cross", "other", 0.0, id="m3_tmall_G4_cross_L6"),
                                                        pytest.param((40, 13, 5, 16, 8, 20, True), "cross", "other", 0.25, id="m3_G2_d16_dropout"),
                                                        pytest.param((9, 3, 31, 12, 32, 20, True), "intra", "x", 0.0, id="m3_G8_forward_only")])
def test_attn_wide_heads_small_d_one_launch_per_direction(lib, case, mode, res_mode, dropout):
    """attn_fwd_wide_kernel / attn_bwd_wide_kernel: the shipped Tmall head geometry (32 x 10 at d = 10) with the head groups looped inside"""
    kc.check_attn_groups_small_d(lib, "cuda", case, mode, res_mode=res_mode, dropout=dropout)


ATTN_EX_CASES = [  # (B, T, S, d, heads, dh, project_out), mode, residual mode, out_scale, softmax_scale
    ((2, 3, 4, 8, 1, 8, True), "intra", "none", 0.5, 0.5),
    ((2, 3, 4, 8, 1, 8, True), "cross", "acc", 0.5, 0.5),
    ((3, 11, 21, 64, 4, 20, True), "intra", "none", 0.5, 10 ** -0.5),     # RAT_m3 at the north-star shape: fast <64, 20>
    ((3, 11, 21, 64, 4, 20, True), "cross", "acc", 0.5, 10 ** -0.5),
    ((5, 6, 14, 40, 4, 20, True), "cross", "other", 1.0, None),           # generic geometry, compile-time dim_head 20
    ((300, 6, 4, 10, 2, 10, True), "intra", "other", 0.25, 0.3),
]


@pytest.mark.parametrize("case,mode,res_mode,out_scale,softmax_scale", ATTN_EX_CASES, ids=str)
def test_attn_ex_fwd_bwd(lib, case, mode, res_mode, out_scale, softmax_scale):
    kc.check_attn_ex(lib, "cuda", case, mode, res_mode, out_scale, softmax_scale)


# round 6: the bf16x3 kernels' 4-head instantiation (attn_fwd3_kernel / attn_bwd3_kernel<.., NH = 4>): RAT_m3's heads / 2 heads of width
# 2 dim_head at the north-star config — every residual form, both phases, many chunks per work-group, a ragged last chunk, P handed to
# pass 2 (L = 11: it fits) and recomputed (L = 21)
@pytest.mark.parametrize("case,mode,res_mode", [((40, 11, 21, 64, 4, 20, True), "intra", "none"), ((40, 11, 21, 64, 4, 20, True), "cross", "acc"),
                                                ((7, 11, 21, 64, 4, 20, True), "cross", "other"), ((3, 5, 31, 64, 4, 20, True), "intra", "acc"),
                                                ((600, 11, 2, 64, 4, 20, True), "cross", "none")], ids=str)
def test_attn_ex_fwd_bwd_bf16x3_four_heads(lib, case, mode, res_mode, knob):
    kc.check_attn_ex(lib, "cuda", case, mode, res_mode, 0.5, 10 ** -0.5, arith="bf16x3")
    knob(lib, "attn_bwd_ph", 0)
    kc.check_attn_ex(lib, "cuda", case, mode, res_mode, 0.5, 10 ** -0.5, arith="bf16x3")


@pytest.mark.parametrize("ntok,d,hidden", [(70, 8, 16), (33, 10, 40), (64, 64, 128), (100000, 64, 128), (5000, 40, 80),
                                           (20011, 10, 40), (20011, 10, 20)])          # (the shipped d = 10 geometries: weights in LDS, one-sweep loads)
def test_ffn_fwd_bwd(lib, ntok, d, hidden):
    kc.check_ffn(lib, "cuda", ntok, d, hidden)


@pytest.mark.parametrize("case,arith", [((2, 3, 4, 8, 2, 4, True), "f32"), ((5, 6, 14, 40, 8, 10, True), "f32"), ((40, 11, 21, 64, 8, 10, True), "f32"),
                                        ((40, 11, 21, 64, 8, 10, True), "bf16x3")], ids=str)
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attention_output_dropout(lib, case, arith, mode):
    kc.check_attn_dropout(lib, "cuda", case, mode, arith=arith)


@pytest.mark.parametrize("d", [40, 48, 56])
@pytest.mark.parametrize("ntok", [77, 100000])
def test_ffn_bf16x3_narrower_layer(lib, d, ntok):
    """(40, 80) — the shipped KKBox feed-forward — / (48, 96) / (56, 112) inside the (64, 128) tiles of the bf16x3 kernels"""
    kc.check_ffn(lib, "cuda", ntok, d, 2 * d, arith="bf16x3")
    kc.check_ffn_res(lib, "cuda", ntok, d, 2 * d, True, arith="bf16x3")
    kc.check_ffn_res(lib, "cuda", ntok, d, 2 * d, False, arith="bf16x3")


@pytest.mark.parametrize("ntok", [64, 1000, 100000])
def test_ffn_fwd_bwd_bf16x3(lib, ntok):
    kc.check_ffn(lib, "cuda", ntok, 64, 128, arith="bf16x3")
    kc.check_ffn_res(lib, "cuda", ntok, 64, 128, True, arith="bf16x3")
    kc.check_ffn_res(lib, "cuda", ntok, 64, 128, False, arith="bf16x3")


@pytest.mark.parametrize("ntok,d,hidden,with_res", [(70, 8, 16, True), (64, 64, 128, True), (100000, 64, 128, True), (100000, 64, 128, False), (5000, 40, 80, True), (999, 16, 32, True)])
def test_ffn_separate_residual(lib, ntok, d, hidden, with_res):
    kc.check_ffn_res(lib, "cuda", ntok, d, hidden, with_res)


@pytest.mark.parametrize("ntok,d,hidden,period", [(4096 * 231, 64, 128, 231), (1000 * 66, 40, 80, 66), (70001, 64, 128, 7), (100, 64, 128, 231), (5, 64, 128, 1)])
def test_ffn_backward_from_compact_gradient_rows(lib, ntok, d, hidden, period):
    kc.check_ffn_rows(lib, "cuda", ntok, d, hidden, period)


@pytest.mark.parametrize("nrows,d,stride_mul,with_add", [(37, 8, 1, False), (100000, 64, 1, True), (8192, 64, 21, False), (4096, 64, 11, True), (1000, 10, 3, True), (300, 400, 2, False)])
def test_layernorm_fwd_bwd(lib, nrows, d, stride_mul, with_add):
    kc.check_layernorm(lib, "cuda", nrows, d, stride_mul, with_add)


@pytest.mark.parametrize("use_bn", [True, False])
def test_bn_relu_colsum(lib, use_bn):
    kc.check_bn_relu(lib, "cuda", 4096, 400, use_bn)
    kc.check_bn_relu(lib, "cuda", 9, 5, use_bn)


@pytest.mark.parametrize("use_bn,act", [(True, "relu"), (False, "relu"), (True, "tanh"), (False, "none")])
@pytest.mark.parametrize("M,N", [(4096, 400), (512, 400), (256, 400), (37, 12), (1500, 64), (5000, 40), (9000, 400), (8, 4)])
def test_bn_act_column_strips(lib, use_bn, act, M, N):
    """every rows-per-thread instantiation (2 / 4 / 8 / 16 and the streaming form above 4096 rows)"""
    kc.check_bn_strip(lib, "cuda", M, N, use_bn, act)
    if act == "relu":
        kc.check_bn_strip_outer(lib, "cuda", M, N, use_bn)


@pytest.mark.parametrize("with_dnn,with_lr", [(True, True), (False, False)])
def test_logit_fwd_bwd(lib, with_dnn, with_lr):
    kc.check_logit(lib, "cuda", 1000, 64, with_dnn, with_lr)


@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_slab_reductions_of_several_layers_in_one_launch(lib, arith):
    kc.check_deferred_reductions(lib, "cuda", arith)


def test_step_begin(lib):
    kc.check_step_begin(lib, "cuda")


def test_l2_sumsq_clip_adam(lib):
    kc.check_optim(lib, "cuda", 1000003)


@pytest.mark.parametrize("M,N", [(4096, 400), (1_000_003, 64), (70_000, 10)])
def test_colsum_long_matrices(lib, M, N):
    """token-sized matrices (composed attention path: bias gradient of to_out) use more row splits than the head's batches"""
    import torch
    from rat_amd import ops
    a = torch.randn(M, N, device="cuda")
    out = torch.empty(N, device="cuda")
    ops.colsum(a, N, out, M, N, lib=lib)
    ref = a.double().sum(0)
    assert float((out.double() - ref).abs().max()) < 1e-4 * max(1.0, M ** 0.5)


@pytest.mark.parametrize("ntok,d,hidden,with_res,add_dy", [(1500, 16, 32, True, False), (777, 10, 20, True, True), (2000, 64, 128, False, False),
                                                            (3000, 40, 80, True, False)])
def test_feed_forward_with_its_dropout_layers(lib, ntok, d, hidden, with_res, add_dy):
    kc.check_ffn_dropout(lib, "cuda", ntok, d, hidden, with_res, add_dy)


@pytest.mark.parametrize("case", [(60, 6, 4, 10, 2, 10, True), (40, 6, 10, 10, 8, 10, True), (33, 5, 3, 16, 4, 10, True), (21, 3, 4, 12, 8, 10, True),
                                  (50, 11, 4, 16, 2, 10, True)], ids=str)
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_generic_kernel_with_one_column_tile(lib, case, mode):
    """embedding_dim <= 16 on the generic kernels: the instantiations with two columns per lane, and — for the shipped MovieLens / Tmall
    geometries (d = 10; 2 or 8 heads) and BASELINE configs[0] (d = 16, 2 heads) — compile-time geometry, where the d(LayerNorm out) GEMM
    (one column tile) splits its contraction over the waves and LayerNorm backward adds the four partial tiles (round 4)"""
    kc.check_attn(lib, "cuda", case, mode)


@pytest.mark.parametrize("ntok,d,hidden", [(3000, 10, 40), (3000, 10, 20), (130, 10, 40)])
def test_ffn_compile_time_geometry_of_the_shipped_d10_configs(lib, ntok, d, hidden):
    kc.check_ffn(lib, "cuda", ntok, d, hidden)
    kc.check_ffn_res(lib, "cuda", ntok, d, hidden, True)
