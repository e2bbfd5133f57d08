"""GPU: the Winograd F(4x2, 3x3) tile convolution (csrc/wino42.hip) against

  * the direct fp32-MFMA kernel (the k-ascending fmaf chain the oracle restates bit for bit),
  * the F(2x2, 3x3) kernel, and
  * a float64 convolution on the CPU;

every epilogue it takes (bias, PReLU, residual + trim, the Dtow store), dense tensors and views into padded
buffers, ragged edges (partial row / column blocks), dead-column blocks, batched tile stacks.  Bound: 3e-5
absolute on outputs of unit scale (measured ~1e-5: F(4, 3)'s constants cost about 3 x the rounding of F(2, 3);
the north-star tolerance is 1e-4 over the whole transform) and a relative bound at other scales."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def P():
    from pseudocylindrical_convolution_amd import PCONV
    return PCONV


def conv(monkeypatch, mode, *args, **kwargs):
    monkeypatch.setenv("PCONV_CONV3X3", mode)
    owner = type("Owner", (), {})()
    return P().tile_conv2d(owner, *args, **kwargs)


def data(tn, cin, h, w, cout, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(tn, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / np.sqrt(cin * 9))
    b = torch.randn(cout, generator=g)
    sl = torch.rand(cout, generator=g)
    return x, wt, b, sl


@pytest.mark.parametrize("cfg", [
    # tn, cin, h, w, cout
    (2, 192, 10, 70, 192), (1, 96, 10, 66, 128), (16, 192, 6, 130, 192), (1, 192, 14, 258, 768), (3, 24, 10, 40, 64),
    (2, 48, 6, 36, 64), (2, 192, 66, 260, 192), (1, 96, 18, 1026, 64), (2, 192, 12, 70, 192), (1, 96, 8, 134, 64), (1, 48, 7, 36, 64),
])
def test_wino42_matches_direct_kernel_and_float64(cfg, hip_backend, monkeypatch):
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg)
    assert P()._native.hip_lib().pconv_wino42_supported(cin, h, w, cout, 0) == 1
    ref64 = torch.nn.functional.conv2d(x.double(), wt.double(), b.double())
    rec = type("Probe", (), {"records": []})()
    monkeypatch.setattr(P(), "conv_probe", rec)
    for slope in (None, sl):
        want = ref64 if slope is None else torch.where(ref64 < 0, ref64 * slope.double().view(1, -1, 1, 1), ref64)
        yd = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope.to(DEV) if slope is not None else None).cpu()
        yw = conv(monkeypatch, "wino42!", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope.to(DEV) if slope is not None else None).cpu()
        assert yw.shape == yd.shape
        ed, ew = (yd.double() - want).abs().max().item(), (yw.double() - want).abs().max().item()
        assert ed < 2e-5 and ew < 3e-5, "direct %g, winograd F(4x2) %g from float64" % (ed, ew)
        assert (yw - yd).abs().max().item() < 3e-5
    assert [r[0].split("<")[0] for r in rec.records] == ["conv_mfma_kernel", "wino42_conv3x3_kernel"] * 2


def test_wino42_dispatch_rules(hip_backend, monkeypatch):
    """default mode "wino42": F(4x2) for 64-multiple couts whose output rows fill the 8-row blocks to 8/9 at least,
    F(2x2) for the rest of what Winograd takes (cin % 24 != 0: F(4x2) does not take it at all), the direct kernel
    below four output rows; "wino42!" (tests, probes) sends every layer the kernel takes to it"""
    rec = type("Probe", (), {"records": []})()
    monkeypatch.setattr(P(), "conv_probe", rec)
    monkeypatch.delenv("PCONV_CONV3X3", raising=False)
    assert P().conv3x3_mode() == "wino42"
    for (cfg, want) in (((1, 32, 10, 66, 64), "wino_conv3x3_kernel"), ((1, 96, 5, 66, 128), "conv_mfma_kernel"),
                        ((1, 96, 10, 66, 96), "wino_conv3x3_kernel"), ((1, 96, 10, 66, 128), "wino42_conv3x3_kernel"),
                        ((1, 192, 10, 66, 192), "wino42_conv3x3_kernel"), ((1, 192, 66, 66, 192), "wino42_conv3x3_kernel"),
                        ((1, 192, 16, 66, 192), "wino_conv3x3_kernel"), ((1, 192, 8, 66, 192), "wino_conv3x3_kernel")):
        x, wt, b, _ = data(*cfg)
        rec.records.clear()
        P().tile_conv2d(type("Owner", (), {})(), x.to(DEV), wt.to(DEV), b.to(DEV), 1)
        assert [r[0].split("<")[0] for r in rec.records] == [want], cfg
    # (r6) a remainder of up to four rows behind whole 8-row blocks: the blocks on F(4x2), the remainder on F(2x2)
    for cfg in ((1, 192, 68, 66, 192), (1, 192, 36, 66, 192), (1, 192, 20, 66, 192), (1, 192, 12, 66, 192), (1, 96, 14, 66, 64)):
        x, wt, b, _ = data(*cfg)
        rec.records.clear()
        P().tile_conv2d(type("Owner", (), {})(), x.to(DEV), wt.to(DEV), b.to(DEV), 1)
        assert [r[0].split("<")[0] for r in rec.records] == ["wino42_conv3x3_kernel", "wino_conv3x3_kernel"], cfg
        assert "rows %d of %d" % ((cfg[2] - 2) // 8 * 8, cfg[2] - 2) in rec.records[0][1]
    monkeypatch.setattr(P(), "WINO_ROW_SPLIT", False)       # the round-5 rule (PCONV_WINO_SPLIT=0)
    for (cfg, want) in (((1, 192, 68, 66, 192), "wino42_conv3x3_kernel"), ((1, 192, 36, 66, 192), "wino_conv3x3_kernel")):
        x, wt, b, _ = data(*cfg)
        rec.records.clear()
        P().tile_conv2d(type("Owner", (), {})(), x.to(DEV), wt.to(DEV), b.to(DEV), 1)
        assert [r[0].split("<")[0] for r in rec.records] == [want], cfg
    monkeypatch.setattr(P(), "WINO_ROW_SPLIT", True)
    monkeypatch.setenv("PCONV_CONV3X3", "wino42!")
    for cfg in ((1, 96, 10, 66, 96), (1, 192, 36, 66, 192)):
        x, wt, b, _ = data(*cfg)
        rec.records.clear()
        P().tile_conv2d(type("Owner", (), {})(), x.to(DEV), wt.to(DEV), b.to(DEV), 1)
        assert rec.records[0][0].split("<")[0] == "wino42_conv3x3_kernel", cfg


@pytest.mark.parametrize("cfg", [(16, 192, 10, 70, 192), (32, 96, 6, 134, 64), (16, 192, 18, 262, 192)])
def test_wino42_epilogue_views_and_dead_columns(cfg, hip_backend, monkeypatch):
    """residual + trim + PReLU, input / residual / output inside padded buffers, per-tile column limits
    (dead 64-column blocks are zeros, trimmed columns are zeros, everything else == the direct kernel's)"""
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=8)
    ho, wo = h - 2, w - 2
    g = torch.Generator().manual_seed(9)
    res = torch.randn(tn, cout, ho, wo, generator=g)
    limit = torch.tensor([wo, 40, 64, 3, 65, 128, wo - 1, 1] * 2, dtype=torch.int32).to(DEV)

    def inside(t, p):
        buf = torch.full((t.shape[0], t.shape[1], t.shape[2] + 2 * p, t.shape[3] + 2 * p), 7.0, device=DEV)
        buf[:, :, p:-p, p:-p] = t.to(DEV)
        return buf[:, :, p:-p, p:-p]

    args = (wt.to(DEV), b.to(DEV), 1, sl.to(DEV), limit, 16)
    for (xin, rin, ring) in ((x.to(DEV), res.to(DEV), 0), (inside(x, 2), inside(res, 2), 2), (inside(x, 1), res.to(DEV), 2)):
        yd = conv(monkeypatch, "direct", xin, *args, residual=rin, trim=True, ring=ring)
        yw = conv(monkeypatch, "wino42!", xin, *args, residual=rin, trim=True, ring=ring)
        assert (yw - yd).abs().max().item() < 3e-5
        for t in range(tn):
            lim = int(limit[t % 16])
            assert yw[t, :, :, lim:].abs().max().item() == 0 if lim < wo else True
        if ring:
            buf = yw._pconv_ring[0]
            assert tuple(buf.shape) == (tn, cout, ho + 2 * ring, wo + 2 * ring)
    yd = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, None, limit, 16)
    yw = conv(monkeypatch, "wino42!", x.to(DEV), wt.to(DEV), b.to(DEV), 1, None, limit, 16)
    assert (yw - yd).abs().max().item() < 3e-5


@pytest.mark.parametrize("cfg", [(16, 192, 10, 70, 768), (2, 96, 10, 134, 192)])
def test_wino42_depth_to_width_store(cfg, hip_backend, monkeypatch):
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=11)
    limit = torch.tensor([w - 2, 40, 64, 3] * 4, dtype=torch.int32).to(DEV)
    for slope in (None, sl.to(DEV)):
        plain = conv(monkeypatch, "wino42!", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16)
        shuffled = P().DtowOp(2, True, 0, False).forward(plain)[0].clone()
        for ring in (0, 2):
            fused = conv(monkeypatch, "wino42!", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16, d2w=True, ring=ring)
            assert tuple(fused.shape) == tuple(shuffled.shape)
            assert torch.equal(fused, shuffled)                     # same arithmetic, another store
        direct = conv(monkeypatch, "direct", x.to(DEV), wt.to(DEV), b.to(DEV), 1, slope, limit, 16, d2w=True)
        assert (direct - shuffled).abs().max().item() < 3e-5


@pytest.mark.parametrize("xscale,wscale", [(1e3, 1.0), (1e-3, 1.0), (1.0, 1e3), (30.0, 30.0)])
def test_wino42_relative_error_at_other_scales(xscale, wscale, hip_backend, monkeypatch):
    """relative to the layer's output scale sqrt(sum_k w^2 x^2) the error against float64 stays under 2e-5 at every
    scale (F(2x2): 1e-5, the direct kernel's 1728-term chain: 2e-5)"""
    tn, cin, h, w, cout = 2, 192, 10, 134, 192
    g = torch.Generator().manual_seed(21)
    x = torch.randn(tn, cin, h, w, generator=g) * xscale
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (wscale / np.sqrt(cin * 9))
    wt[::4] *= 100.0
    wt[:, ::3] *= 0.01
    b = torch.zeros(cout)
    ref64 = torch.nn.functional.conv2d(x.double(), wt.double())
    scale = torch.nn.functional.conv2d(x.double() ** 2, wt.double() ** 2).sqrt()
    yw = conv(monkeypatch, "wino42!", x.to(DEV), wt.to(DEV), b.to(DEV), 1).cpu().double()
    rw = ((yw - ref64).abs() / scale).max().item()
    assert rw < 2e-5, "relative error of F(4x2, 3x3): %g" % rw
    assert torch.isfinite(yw).all()


@pytest.mark.parametrize("cfg", [(2, 192, 68, 132, 192), (16, 192, 36, 70, 192), (3, 96, 20, 262, 64), (16, 192, 12, 134, 192)])
def test_row_split_equals_the_direct_kernel(cfg, hip_backend, monkeypatch):
    """(r6) default dispatch of a layer whose output rows are whole 8-row blocks plus a remainder of two or four rows (66, 34,
    18, 10): F(4x2) on the blocks, F(2x2) on the remainder, two launches over row views of the same tensors.
    Against the direct kernel and float64; with residual + trim + PReLU + column limits and with input / residual /
    output inside padded buffers (the row views then start in the middle of a padded buffer)."""
    tn, cin, h, w, cout = cfg
    x, wt, b, sl = data(*cfg, seed=12)
    ho, wo = h - 2, w - 2
    res = torch.randn(tn, cout, ho, wo, generator=torch.Generator().manual_seed(13))
    limit = torch.tensor([wo, 40, 64, 3, 65, 128, wo - 1, 1] * 2, dtype=torch.int32).to(DEV)
    rec = type("Probe", (), {"records": []})()
    monkeypatch.setattr(P(), "conv_probe", rec)
    ref64 = torch.nn.functional.conv2d(x.double(), wt.double(), b.double())
    y0 = conv(monkeypatch, "wino42", x.to(DEV), wt.to(DEV), b.to(DEV), 1).cpu()
    assert [r[0] for r in rec.records] == ["wino42_conv3x3_kernel", "wino_conv3x3_kernel"]
    assert (y0.double() - ref64).abs().max().item() < 3e-5

    def inside(t, p):
        buf = torch.full((t.shape[0], t.shape[1], t.shape[2] + 2 * p, t.shape[3] + 2 * p), 7.0, device=DEV)
        buf[:, :, p:-p, p:-p] = t.to(DEV)
        return buf[:, :, p:-p, p:-p]

    args = (wt.to(DEV), b.to(DEV), 1, sl.to(DEV), limit, 16)
    for (xin, rin, ring) in ((x.to(DEV), res.to(DEV), 0), (inside(x, 2), inside(res, 2), 2), (inside(x, 1), res.to(DEV), 2)):
        yd = conv(monkeypatch, "direct", xin, *args, residual=rin, trim=True, ring=ring)
        rec.records.clear()
        yw = conv(monkeypatch, "wino42", xin, *args, residual=rin, trim=True, ring=ring)
        assert [r[0] for r in rec.records] == ["wino42_conv3x3_kernel", "wino_conv3x3_kernel"]
        assert yw.shape == yd.shape and (yw - yd).abs().max().item() < 3e-5
        if ring:
            buf, r = yw._pconv_ring
            assert buf.shape[2] == ho + 2 * r and torch.equal(buf[:, :, r:-r, r:-r], yw)
