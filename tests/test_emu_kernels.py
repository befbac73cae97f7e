"""CPU checks of the kernel SOURCES through the host-emulation build (tests/emu): the same .hip files compiled
with g++ -DRAT_EMU, one OS thread per GPU thread.  They pin index arithmetic / LDS layouts / the MFMA lane maps
against the oracle on tiny shapes without a GPU; tests/test_gpu_kernels.py (-m gpu) runs the same checks, and
bigger ones, on the real HIP build."""
import os
import sys

import pytest
from conftest import gpu_twin, twin

import kernel_cases as kc
from rat_amd._lib import RatLib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.fixture(scope="session")
def emu():
    import build_emu
    return RatLib(build_emu.build())


@pytest.mark.parametrize("d", [8, 10, 64])           # 64: the wave-per-row scatter kernel (sequence field with 3 ids included)
def test_gather_fwd_bwd(emu, d):
    kc.check_gather(emu, "cpu", d)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_sgemm(emu, ta, tb):
    kc.check_sgemm(emu, "cpu", ta, tb)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_sgemm_bf16x3(emu, ta, tb):
    kc.check_sgemm(emu, "cpu", ta, tb, M=72, N=40, K=100, arith="bf16x3")      # ragged last k tile, partial n tile


def test_sgemm_split_k(emu):
    """few output tiles, long K: the k range is split across work-groups and reduced in slice order"""
    assert emu.size("rat_sgemm_workspace", 40, 33, 700) > 0
    kc.check_sgemm(emu, "cpu", 1, 0, 40, 33, 700)
    kc.check_sgemm(emu, "cpu", 0, 1, 3, 70, 515)


ATTN_CASES = [  # B, T, S, d, heads, dh, project_out
    (2, 3, 4, 8, 2, 4, True),
    (1, 4, 5, 10, 2, 10, True),
    (2, 2, 3, 8, 1, 8, False),
    (1, 3, 4, 16, 8, 10, True),      # served by the compiled fast shape <TD=16, TDH=10>
    (1, 2, 3, 64, 8, 10, True),      # fast shape <TD=64, TDH=10> (split-K dXn path)
    (1, 2, 19, 16, 8, 10, True),     # intra length 19 -> two key/query tiles in the MFMA attention core
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=str)
@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_fwd_bwd(emu, case, mode):
    kc.check_attn(emu, "cpu", case, mode)


@pytest.fixture()
def two_blocks(emu, knob):
    knob(emu, "max_blocks", 2)


@pytest.mark.parametrize("mode", ["intra", "cross"])
def test_attn_fwd_bwd_bf16x3(emu, mode, two_blocks):
    """the split-operand bf16 MFMA kernels (plane images in LDS, transposed block reads, pre-split weight fragments) through the
    emulated v_mfma_f32_16x16x32_bf16 / ds_read_b64_tr_b16.  knob max_blocks = 2: two work-groups loop over 4-5 chunks each — persistent weight-gradient accumulators, double-buffered row maps, load-ahead, a ragged
    last chunk."""
    kc.check_attn(emu, "cpu", (2, 6, 21, 64, 8, 10, True), mode, arith="bf16x3")


@pytest.mark.parametrize("case,mode", [pytest.param((1, 3, 14, 40, 8, 10, True), "intra", id="d40_L14"), pytest.param((2, 6, 5, 40, 8, 10, True), "cross", id="d40_L6"),
                                       twin((1, 2, 9, 56, 8, 10, True), "intra", id="d56_L9")])
def test_attn_fwd_bwd_bf16x3_narrower_embedding(emu, case, mode, two_blocks):
    """embedding_dim 40 (the shipped KKBox config) / 48 / 56 inside the 64-wide tiles of the bf16x3 kernels: 160-byte rows, pieces beyond
    d zero, LayerNorm over d columns, zero-padded weight planes, d-wide gradient slabs"""
    kc.check_attn(emu, "cpu", case, mode, arith="bf16x3")


def test_attn_bwd_bf16x3_probabilities_handed_to_pass_two(emu, two_blocks, knob):
    """for L <= 12 pass 1 of attn_bwd3_kernel leaves P in LDS and pass 2 reads it (the default; knob attn_bwd_ph = 0 recomputes) — both forms"""
    kc.check_attn(emu, "cpu", (2, 3, 7, 64, 8, 10, True), "intra", arith="bf16x3")
    knob(emu, "attn_bwd_ph", 0)
    kc.check_attn(emu, "cpu", (3, 11, 4, 64, 8, 10, True), "cross", arith="bf16x3")


@pytest.mark.parametrize("case,mode", [pytest.param((2, 6, 21, 64, 8, 10, True), "intra", id="L21_two_tiles"),
                                       pytest.param((3, 11, 4, 64, 8, 10, True), "cross", id="L11_one_tile"),
                                       twin((1, 31, 2, 64, 8, 10, True), "cross", id="L31")])
def test_attn_bwd_bf16x3_matrix_pipe_core(emu, case, mode, two_blocks, knob):
    """attn_bwd3_kernel<.., MC>: the backward core on v_mfma_f32_16x16x4_f32 (one wave per (sequence, head) pair, P / dS through a
    wave-private LDS tile) — forced on by the knob at lengths the host would leave to the VALU passes: two 16-row tiles with a ragged
    second one (L = 21), one ragged tile (L = 11)"""
    knob(emu, "attn_bwd_core_mfma", 1)
    kc.check_attn(emu, "cpu", case, mode, arith="bf16x3")


def test_attn_bwd_bf16x3_matrix_pipe_core_three_tiles(emu, two_blocks):
    """41 tokens: one sequence per chunk, the backward core's key-tile-inner form on three 16-row tiles (the host's rule from 40 tokens on),
    the forward core on three tiles as well; two work-groups over three chunks"""
    kc.check_attn(emu, "cpu", (1, 3, 41, 64, 8, 10, True), "intra", arith="bf16x3")


def test_attn_narrower_embedding_with_queries_and_dropout(emu, two_blocks):
    kc.check_attn_queries(emu, "cpu", (2, 3, 7, 40, 8, 10, True), "intra", nq=1, arith="bf16x3")
    kc.check_attn_dropout(emu, "cpu", (1, 3, 5, 40, 8, 10, True), "cross", arith="bf16x3")


@pytest.mark.parametrize("case,mode", [pytest.param((2, 3, 21, 64, 8, 10, True), "intra", id="L21_two_tiles"),
                                       pytest.param((3, 11, 4, 64, 8, 10, True), "cross", id="L11_one_tile"),
                                       twin((1, 31, 2, 64, 8, 10, True), "cross", id="L31"),
                                       pytest.param((1, 3, 41, 64, 8, 10, True), "intra", id="L41_three_tiles")])
def test_attn_fwd_exact_fp32_matrix_pipe_core(emu, case, mode, two_blocks, knob):
    """attn_fwd3_kernel<.., MCF>: the forward core on v_mfma_f32_16x16x4_f32 (S^T = K Q^T, the row softmax in the accumulators, which then
    ARE the B operand of O^T = V^T P^T) — forced on by the knob at lengths the host would leave to the VALU loop: two 16-row tiles with a
    ragged second one (L = 21), one ragged tile (L = 11); L = 31 is where the host selects it by itself"""
    knob(emu, "attn_fwd_core_mfma", 2)
    kc.check_attn(emu, "cpu", case, mode, arith="bf16x3")


@pytest.mark.parametrize("case,mode,nq,arith", [pytest.param((2, 6, 21, 64, 8, 10, True), "intra", 1, "bf16x3", id="b3_intra_L21"),
                                                twin((3, 11, 4, 64, 8, 10, True), "cross", 1, "bf16x3", id="b3_cross_L11"),
                                                pytest.param((1, 3, 7, 64, 8, 10, True), "intra", 3, "bf16x3", id="b3_three_queries"),
                                                pytest.param((2, 3, 4, 8, 2, 4, True), "intra", 1, "f32", id="generic_ignores"),
                                                twin((1, 2, 5, 64, 8, 10, True), "cross", 1, "f32", id="fast_f32_ignores")])
def test_attn_with_a_subset_of_query_positions(emu, case, mode, nq, arith, two_blocks):
    """RatSeqMap.queries: the bf16x3 kernels skip the other queries, the exact-fp32 kernels compute them all (both within the contract)"""
    kc.check_attn_queries(emu, "cpu", case, mode, nq=nq, arith=arith)


@pytest.mark.parametrize("case,arith", [((2, 3, 4, 8, 2, 4, True), "f32"), ((1, 2, 5, 64, 8, 10, True), "f32"), ((1, 3, 5, 64, 8, 10, True), "bf16x3")],
                         ids=["generic", "fast_f32", "bf16x3"])
def test_attention_output_dropout(emu, case, arith):
    kc.check_attn_dropout(emu, "cpu", case, "intra", arith=arith)
    kc.check_attn_dropout(emu, "cpu", case, "cross", arith=arith)


def test_persistent_kernels_loop_over_several_chunks(emu, two_blocks):
    """exact-fp32 fast kernels and the FFN with every work-group looping over several chunks (knob max_blocks = 2)"""
    kc.check_attn(emu, "cpu", (2, 5, 9, 64, 8, 10, True), "intra")
    kc.check_ffn(emu, "cpu", 200, 64, 128)
    kc.check_ffn(emu, "cpu", 200, 64, 128, arith="bf16x3")


# (1, 70, 1, 10) and (3, 49, 2, 10): dim_head 10 with 48+ tokens — the forward on the matrix pipe (core_fwd_mfma_kernel): five key tiles = an odd
# count (the second tile of the last key block is the spare zero tile), a ragged last tile; several pairs per work-group with RAT_MAX_BLOCKS
@pytest.mark.parametrize("nseq,L,heads,dh,softmax_scale", [(2, 5, 2, 4, None), (1, 70, 1, 10, None), (2, 33, 2, 7, 0.3), (3, 49, 2, 10, 0.4)])
def test_attn_core_fwd_bwd(emu, nseq, L, heads, dh, softmax_scale, knob):
    """dim_head 10 with 48+ tokens: the backward is the hybrid kernel (core_bwd_hybrid_kernel: dQ on the matrix pipe beside dK / dV on the
    VALU); the attn_bwd_core_mfma knob's value 0 keeps the two VALU passes — both forms"""
    kc.check_attn_core(emu, "cpu", nseq, L, heads, dh, softmax_scale)
    if dh == 10 and L >= 48:
        knob(emu, "attn_bwd_core_mfma", 0)
        kc.check_attn_core(emu, "cpu", nseq, L, heads, dh, softmax_scale)


def test_attn_core_matrix_pipe_forward_several_pairs_per_work_group(emu, knob):
    """core_fwd_mfma_kernel with fewer work-groups than (sequence, head) pairs: the K / V tiles are re-staged per pair"""
    knob(emu, "max_blocks", 1)                                   # 8 work-groups for 10 pairs (backward: 4 — every one re-stages its tiles)
    kc.check_attn_core(emu, "cpu", 5, 50, 2, 10, None)


@pytest.mark.parametrize("B,T,S,heads,dh", [(2, 3, 4, 2, 4), (1, 5, 3, 3, 10)])
def test_attn_core_strided(emu, B, T, S, heads, dh):
    kc.check_attn_core_strided(emu, "cpu", B, T, S, heads, dh)


@pytest.mark.parametrize("case,mode,res_mode,dropout", [pytest.param((1, 3, 5, 64, 16, 10, True), "intra", "x", 0.0, id="G2_intra"),
                                                        twin((1, 4, 3, 64, 32, 10, True), "cross", "other", 0.25, id="G4_cross_dropout")])
def test_attn_wide_heads_group_loop(emu, case, mode, res_mode, dropout, two_blocks):
    """rat_attn_fwd_groups (attn_fwd3_kernel<GRP>): every head group of a chunk inside one launch"""
    kc.check_attn_groups(emu, "cpu", case, mode, res_mode=res_mode, dropout=dropout)


@pytest.mark.parametrize("case,mode,res_mode,dropout", [pytest.param((2, 3, 5, 10, 32, 10, True), "intra", "x", 0.0, id="tmall_G4_intra"),
                                                        pytest.param((1, 4, 3, 10, 16, 10, True), "cross", "other", 0.25, id="G2_cross_dropout"),
                                                        twin((1, 3, 2, 16, 24, 10, True), "intra", "x", 0.0, id="G3_d16"),
                                                        pytest.param((2, 3, 5, 10, 16, 20, True), "cross", "other", 0.2, id="m3_tmall_G4_4x20")])
def test_attn_wide_heads_small_d_one_launch_per_direction(emu, case, mode, res_mode, dropout, two_blocks):
    """attn_fwd_wide_kernel / attn_bwd_wide_kernel: the shipped Tmall head geometry (32 x 10 at d = 10) with the head groups looped inside"""
    kc.check_attn_groups_small_d(emu, "cpu", case, mode, res_mode=res_mode, dropout=dropout)


ATTN_EX_CASES = [  # (B, T, S, d, heads, dh, project_out), mode, residual mode, out_scale, softmax_scale
    ((2, 3, 4, 8, 1, 8, True), "intra", "none", 0.5, 0.5),
    ((2, 3, 4, 8, 1, 8, True), "cross", "acc", 0.5, 0.5),
    ((1, 2, 3, 64, 4, 20, True), "intra", "none", 0.5, 10 ** -0.5),       # RAT_m3 at the north-star widths: fast <64, 20>
    ((1, 2, 3, 64, 4, 20, True), "cross", "acc", 0.5, 10 ** -0.5),
    ((1, 3, 4, 16, 4, 20, True), "cross", "other", 1.0, None),            # generic geometry, compile-time dim_head 20
]


@pytest.mark.parametrize("case,mode,res_mode,out_scale,softmax_scale", ATTN_EX_CASES, ids=str)
def test_attn_ex_fwd_bwd(emu, case, mode, res_mode, out_scale, softmax_scale):
    kc.check_attn_ex(emu, "cpu", case, mode, res_mode, out_scale, softmax_scale)


@pytest.mark.parametrize("case,mode,res_mode", [((2, 3, 21, 64, 4, 20, True), "intra", "none"), ((3, 11, 4, 64, 4, 20, True), "cross", "acc")], ids=str)
def test_attn_ex_fwd_bwd_bf16x3_four_heads(emu, case, mode, res_mode, two_blocks):
    """round 6: attn_fwd3_kernel / attn_bwd3_kernel<.., NH = 4> (RAT_m3 at the north-star config): two work-groups over several chunks with a
    ragged last one; L = 21 recomputes P in pass 2, L = 11 hands it over"""
    kc.check_attn_ex(emu, "cpu", case, mode, res_mode, 0.5, 10 ** -0.5, arith="bf16x3")


def test_ffn_bf16x3_narrower_layer(emu, two_blocks):
    """(40, 80) — the shipped KKBox feed-forward — inside the (64, 128) tiles of the bf16x3 kernels: zero-padded weight planes / staged
    weights, guarded token fragments, (hidden, d)-shaped gradient slabs"""
    kc.check_ffn(emu, "cpu", 150, 40, 80, arith="bf16x3")
    kc.check_ffn_res(emu, "cpu", 77, 40, 80, True, arith="bf16x3")


def test_ffn_backward_from_compact_gradient_rows(emu, two_blocks):
    """the last block's feed-forward backward: only one token row per sample carries a gradient (rat_ffn_bwd_res_rows)"""
    kc.check_ffn_rows(emu, "cpu", 231, 64, 128, 21)            # 11 rows of 231 tokens; several chunks per work-group
    kc.check_ffn_rows(emu, "cpu", 150, 40, 80, 7)              # 150 = 21 * 7 + 3: the last row's period is cut short


@pytest.mark.parametrize("ntok,d,hidden", [(70, 8, 16), (33, 10, 40), (64, 64, 128), (77, 64, 128), (45, 16, 32)])
def test_ffn_fwd_bwd(emu, ntok, d, hidden):
    kc.check_ffn(emu, "cpu", ntok, d, hidden)


@pytest.mark.parametrize("ntok,d,hidden,with_res", [(70, 8, 16, True), (64, 64, 128, True), (77, 64, 128, False), (45, 16, 32, True)])
def test_ffn_separate_residual(emu, ntok, d, hidden, with_res):
    kc.check_ffn_res(emu, "cpu", ntok, d, hidden, with_res)


@pytest.mark.parametrize("nrows,d,stride_mul,with_add", [(37, 8, 1, False), (50, 64, 1, True), (21, 64, 21, False), (19, 10, 3, True), (5, 100, 2, False)])
def test_layernorm_fwd_bwd(emu, nrows, d, stride_mul, with_add):
    kc.check_layernorm(emu, "cpu", nrows, d, stride_mul, with_add)


@pytest.mark.parametrize("use_bn", [True, False])
def test_bn_relu_colsum(emu, use_bn):
    kc.check_bn_relu(emu, "cpu", 37, 40, use_bn)
    kc.check_bn_relu(emu, "cpu", 9, 5, use_bn)


@pytest.mark.parametrize("use_bn,act", [(True, "relu"), (False, "relu"), (True, "tanh")])
def test_bn_act_column_strips(emu, use_bn, act):
    kc.check_bn_strip(emu, "cpu", 37, 40, use_bn, act)         # one row per thread at most, 5 column groups on 8 "XCDs"
    kc.check_bn_strip(emu, "cpu", 600, 12, use_bn, act)        # three rows per thread (RPT 4), a half-filled last column group
    if act == "relu":
        kc.check_bn_strip_outer(emu, "cpu", 300, 20, use_bn)
    if not use_bn:                                             # (two-row batch statistics are all cancellation: no float64 comparison)
        kc.check_bn_strip(emu, "cpu", 2, 4, use_bn, act)       # the smallest matrices the strips accept
        kc.check_bn_strip(emu, "cpu", 1, 8, use_bn, act)


@pytest.mark.parametrize("with_dnn,with_lr", [(True, True), (False, False)])
def test_logit_fwd_bwd(emu, with_dnn, with_lr):
    kc.check_logit(emu, "cpu", 9, 8, with_dnn, with_lr)


def test_slab_reductions_of_several_layers_in_one_launch(emu, two_blocks):
    kc.check_deferred_reductions(emu, "cpu")


def test_step_begin(emu):
    kc.check_step_begin(emu, "cpu")


def test_l2_sumsq_clip_adam(emu):
    kc.check_optim(emu, "cpu", 777)


@pytest.mark.parametrize("ntok,d,hidden,with_res,add_dy", [(150, 16, 32, True, False), (77, 10, 20, True, True), twin(200, 64, 128, False, False)])
def test_feed_forward_with_its_dropout_layers(emu, ntok, d, hidden, with_res, add_dy):
    kc.check_ffn_dropout(emu, "cpu", ntok, d, hidden, with_res, add_dy)


@pytest.mark.parametrize("case", [(2, 3, 4, 10, 2, 10, True), twin((2, 4, 9, 10, 8, 10, True))], ids=str)
def test_attn_generic_kernel_with_one_column_tile(emu, two_blocks, case):
    kc.check_attn(emu, "cpu", case, "intra")
    kc.check_attn(emu, "cpu", case, "cross")


@pytest.mark.parametrize("d,hidden", [(10, 40), twin(10, 20)])
def test_ffn_compile_time_geometry_of_the_shipped_d10_configs(emu, d, hidden):
    kc.check_ffn(emu, "cpu", 150, d, hidden)
