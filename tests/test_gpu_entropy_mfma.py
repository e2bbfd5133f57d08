"""GPU: the encoder's hidden layers on the fp32 matrix cores (csrc/entropy_mfma.hip) write the very streams of the
vector kernel (PCONV_EE_BULK=valu) -- lane-class sub-GEMMs folded in the butterfly's order are the published
summation order (entropy.hip), not a new one: no oracle, per-op or step kernel changed.  The whole-codec
comparisons against the oracle (tests/test_gpu_codec_vs_oracle.py) run on the matrix-core form by default."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ent(seed=7):
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    torch.manual_seed(1234)
    enc = PC.PseudoEncoder(56, 0)
    g = torch.Generator().manual_seed(seed)
    sd = {k: torch.randn(v.shape, generator=g) * 0.05 for k, v in enc.ent.state_dict().items()}
    enc.ent.load_state_dict(sd)
    return enc.ent


def _encode(ent, sym, h, w, n, mode, monkeypatch, ranges=None):
    from pseudocylindrical_convolution_amd.engine import EntropyEngine
    monkeypatch.setenv("PCONV_EE_BULK", mode)
    if ranges is not None:
        monkeypatch.setenv("PCONV_ENGINE_ENCODE_RANGES", str(ranges))
    return EntropyEngine(ent, h, w, n, "cuda:0").encode(sym)


@pytest.mark.parametrize("h,w,n", [(2, 64, 1), (4, 128, 2), (8, 256, 1), (16, 512, 1), (2, 80, 3), (6, 96, 1)])
def test_matrix_core_encoder_writes_the_vector_kernels_streams(hip_backend, monkeypatch, h, w, n):
    """rows per tile 2 .. 16 (block shapes 2 x 64, 4 x 32), widths that are not multiples of a block, frames in
    lock-step, and a row count (6) whose blocks are 2 rows high"""
    ent = _ent()
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(3 + h)).float().cuda()
    sym = ent.fill(sym).contiguous()
    a = _encode(ent, sym, h, w, n, "valu", monkeypatch)
    b = _encode(ent, sym, h, w, n, "mfma", monkeypatch)
    assert a == b
    assert all(len(s) > 16 for s in b)


def test_matrix_core_encoder_in_one_piece_and_in_step_ranges(hip_backend, monkeypatch):
    """the last group of a call is coded in step ranges (engine.cpp): the matrix-core kernel filters its stores by
    range and skips blocks outside it; one piece and eight pieces give the same streams as the vector kernel"""
    ent = _ent(11)
    h, w, n = 8, 256, 2
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(5)).float().cuda()
    sym = ent.fill(sym).contiguous()
    ref = _encode(ent, sym, h, w, n, "valu", monkeypatch, ranges=1)
    for r in (1, 3, 8):
        assert _encode(ent, sym, h, w, n, "mfma", monkeypatch, ranges=r) == ref


def test_odd_row_counts_take_one_row_blocks(hip_backend, monkeypatch):
    """three rows per tile: blocks of one row x 64 columns (one row per wave); with two rows per wave asked for
    (PCONV_EE_MFMA_NT=2) the block shape still falls back to one"""
    ent = _ent()
    h, w, n = 3, 64, 1
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(9)).float().cuda()
    sym = ent.fill(sym).contiguous()
    ref = _encode(ent, sym, h, w, n, "valu", monkeypatch)
    assert _encode(ent, sym, h, w, n, "mfma", monkeypatch) == ref
    monkeypatch.setenv("PCONV_EE_MFMA_NT", "2")
    monkeypatch.setenv("PCONV_EE_MFMA_WSRC", "ring")
    assert _encode(ent, sym, h, w, n, "mfma", monkeypatch) == ref


@pytest.mark.parametrize("nt,waves,wsrc", [(1, 4, "direct"), (1, 4, "ring"), (2, 4, "ring"), (1, 8, "direct"), (2, 8, "ring")])
def test_measured_variants_of_the_matrix_core_kernel_agree(hip_backend, monkeypatch, nt, waves, wsrc):
    """rows per wave x waves per workgroup x where the weight fragments come from (PCONV_EE_MFMA_NT / _WAVES / _WSRC:
    the variants of profiles/round5_entropy_mfma_variants.txt) all write the vector kernel's streams"""
    ent = _ent(13)
    h, w, n = 8, 192, 1
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(17)).float().cuda()
    sym = ent.fill(sym).contiguous()
    ref = _encode(ent, sym, h, w, n, "valu", monkeypatch, ranges=2)
    monkeypatch.setenv("PCONV_EE_MFMA_FORM", "16x4")
    monkeypatch.setenv("PCONV_EE_MFMA_NT", str(nt))
    monkeypatch.setenv("PCONV_EE_MFMA_WAVES", str(waves))
    monkeypatch.setenv("PCONV_EE_MFMA_WSRC", wsrc)
    assert _encode(ent, sym, h, w, n, "mfma", monkeypatch, ranges=2) == ref


@pytest.mark.parametrize("h,w,n,waves,ranges", [(8, 192, 1, 4, 2), (16, 512, 1, 4, 1), (3, 64, 1, 4, 1), (2, 80, 3, 4, 3),
                                                (8, 256, 2, 8, 1)])
def test_four_block_form_of_the_matrix_core_kernel_agrees(hip_backend, monkeypatch, h, w, n, waves, ranges):
    """four lane classes per instruction (v_mfma_f32_16x16x1_4b_f32, the default form of the hidden layers; the
    16 x 16 x 4 form is PCONV_EE_MFMA_FORM=16x4), the first two butterfly levels inside a lane: the same streams"""
    ent = _ent(19)
    sym = torch.randint(0, 8, (16 * n, 14, h, w), generator=torch.Generator().manual_seed(23 + h)).float().cuda()
    sym = ent.fill(sym).contiguous()
    ref = _encode(ent, sym, h, w, n, "valu", monkeypatch, ranges=ranges)
    monkeypatch.setenv("PCONV_EE_MFMA_WAVES", str(waves))
    monkeypatch.setenv("PCONV_EE_MFMA_FORM", "4b")
    assert _encode(ent, sym, h, w, n, "mfma", monkeypatch, ranges=ranges) == ref
    monkeypatch.setenv("PCONV_EE_MFMA_FORM", "16x4")
    assert _encode(ent, sym, h, w, n, "mfma", monkeypatch, ranges=ranges) == ref


@pytest.mark.parametrize("h,w", [(4, 65), (2, 81)])
def test_odd_widths_keep_the_vector_kernel(hip_backend, monkeypatch, h, w):
    """a padded row of an odd width ends on half a 16-byte piece of the matrix-core kernel's patch loads: the engine
    takes the vector kernel for such shapes (csrc/engine.cpp; only direct pconv_ee_create users meet them -- the
    codec's symbol planes are Dtow(2) outputs).  Same streams as PCONV_EE_BULK=valu, and the decoder (step kernel)
    reads them back."""
    from pseudocylindrical_convolution_amd.engine import EntropyEngine
    ent = _ent(29)
    sym = torch.randint(0, 8, (16, 14, h, w), generator=torch.Generator().manual_seed(31 + w)).float().cuda()
    sym = ent.fill(sym).contiguous()
    ref = _encode(ent, sym, h, w, 1, "valu", monkeypatch)
    out = _encode(ent, sym, h, w, 1, "mfma", monkeypatch)
    assert out == ref
    back = EntropyEngine(ent, h, w, 1, "cuda:0").decode(out)
    assert torch.equal(back, sym)


def test_packed_rows_refuse_a_label_outside_the_alphabet(hip_backend, monkeypatch):
    """a symbol such as 256 must not be folded to 0 by the packed 16-byte rows: the coder answers as it does for the
    int32 rows (include/pconv_coder.h: symbol out of range)"""
    from pseudocylindrical_convolution_amd._native import PconvError
    ent = _ent(37)
    h, w = 2, 64
    sym = torch.randint(0, 8, (16, 14, h, w), generator=torch.Generator().manual_seed(41)).float().cuda()
    sym = ent.fill(sym).contiguous()
    sym[5, 3, 1, 2] = 256.0
    with pytest.raises(PconvError):
        _encode(ent, sym, h, w, 1, "mfma", monkeypatch)
