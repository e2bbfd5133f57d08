"""GPU: the device side of the reference's frame I/O (pseudo_codec.py:215-221) and the double-buffered host <-> HBM
pipe bench.py's step starts and ends at.  The reference statements ARE numpy one-liners, so they are the oracle:
    img2tensor: torch.from_numpy(img.transpose(2,0,1).astype(np.float32)) / 255.
    tensor2img: (data[0] * 255.).to('cpu').numpy().transpose(1,2,0).astype(np.uint8)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_img2tensor_on_the_device_is_the_reference_division(hip_backend):
    from pseudocylindrical_convolution_amd import PCONV
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (3, 64, 48, 3), generator=g, dtype=torch.uint8)
    img[0, 0, :16, 0] = torch.arange(0, 256, 16, dtype=torch.uint8)          # every 16th byte value at least once
    img[1].view(-1)[:256] = torch.arange(256, dtype=torch.uint8)             # ... and all 256 of them
    want = torch.from_numpy(img.numpy().transpose(0, 3, 1, 2).astype(np.float32)) / 255.
    got = PCONV.frames_u8_to_f32(img.cuda()).cpu()
    assert got.shape == want.shape and torch.equal(got, want)


def test_tensor2img_on_the_device_is_numpys_cast(hip_backend):
    """values inside [0, 1], on the edges, and past them (ClipData leaks with slope 0.01: 1.004 * 255 = 256.02 wraps
    to 0 in numpy's float32 -> uint8 cast; small negatives truncate to 0)"""
    from pseudocylindrical_convolution_amd import PCONV
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 32, 64, generator=g)
    x[0, 0, 0, :8] = torch.tensor([0.0, 1.0, 1.004, -0.001, 0.99999, 254.5 / 255, 1.0039, -0.0038])
    x[1, 2, 5, :4] = torch.tensor([0.5, 127.999 / 255, 128.0 / 255, 1.0 - 1e-7])
    want = (x * 255.).numpy().transpose(0, 2, 3, 1).astype(np.uint8)
    got = PCONV.frames_f32_to_u8(x.cuda()).cpu().numpy()
    assert got.shape == want.shape and (got == want).all()


def test_round_trip_of_an_image_is_the_identity(hip_backend):
    from pseudocylindrical_convolution_amd import PCONV
    img = torch.randint(0, 256, (1, 256, 512, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8).cuda()
    back = PCONV.frames_f32_to_u8(PCONV.frames_u8_to_f32(img))
    # u8 / 255 * 255 truncated: numpy gives the same bytes except where the product lands just below the integer
    want = ((img.cpu().numpy().astype(np.float32) / 255.) * np.float32(255.)).astype(np.uint8)
    assert (back.cpu().numpy() == want).all()


def test_shapes_the_kernels_refuse(hip_backend):
    from pseudocylindrical_convolution_amd import PCONV
    from pseudocylindrical_convolution_amd._native import PconvError
    with pytest.raises(PconvError):
        PCONV.frames_u8_to_f32(torch.zeros((1, 4, 6, 3), dtype=torch.uint8).cuda())       # width % 4
    with pytest.raises(PconvError):
        PCONV.frames_u8_to_f32(torch.zeros((1, 4, 8, 4), dtype=torch.uint8).cuda())       # not 3 channels
    with pytest.raises(PconvError):
        PCONV.frames_f32_to_u8(torch.zeros((1, 3, 4, 8), dtype=torch.float64).cuda())


def test_frame_pipe_double_buffers_batches_between_host_and_hbm(hip_backend):
    """three batches through the pipe as bench.py's step drives it: upload of batch k + 1 queued while batch k is
    in use, downloads into alternating pinned buffers; what arrives is img2tensor / tensor2img of what was sent"""
    from pseudocylindrical_convolution_amd.engine import FramePipe
    n, h, w = 2, 64, 128
    pipe = FramePipe(n, h, w, "cuda:0")
    g = torch.Generator().manual_seed(4)
    batches = [torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(3)]
    pipe.prefetch(batches[0], 0)
    outs = []
    for k in range(3):
        slot = k & 1
        frames = pipe.take(slot)
        if k + 1 < 3:
            pipe.prefetch(batches[k + 1], slot ^ 1)
        want = torch.from_numpy(batches[k].numpy().transpose(0, 3, 1, 2).astype(np.float32)) / 255.
        assert torch.equal(frames.cpu(), want)
        rec = (frames * 0.5 + 0.25).contiguous()                 # any "reconstruction"
        host = pipe.give(rec, slot)
        outs.append((slot, (rec * 255.).cpu().numpy().transpose(0, 2, 3, 1).astype(np.uint8)))
        if k >= 1:                                               # the previous batch's download, checked a step late
            pslot, pwant = outs[k - 1]
            if pslot != slot:
                assert (pipe.wait(pslot).numpy() == pwant).all()
    assert (pipe.wait(outs[-1][0]).numpy() == outs[-1][1]).all()
    assert host.is_pinned()
