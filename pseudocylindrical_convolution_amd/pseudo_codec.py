"""360-degree image codec driver (reference: pseudo_codec.py:27-356).

Inference-time entropy model (EntEncoder / EntDecoder wavefront loops), the
end-to-end PseudoEncoder / PseudoDecoder, evaluation and the command line.
Differences from the reference, all additive:
  * any ERP size with height % 256 == 0 and width % 16 == 0 (the reference
    hard-codes 512x1024 latents, pseudo_codec.py:206,209,229-234); the defaults
    reproduce the reference exactly;
  * images are read/written with PIL (cv2 is not required) in the reference's BGR
    channel order, so its checkpoints stay valid.
Module / parameter names are the reference's, so `{idx}_encoder.pt`,
`{idx}_decoder.pt` and `{idx}_ent.pt` load with strict=True (pseudo_codec.py:223-227).
"""
import argparse
import math
import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .PCONV_operator import (DExtract2, DExtract2Batch, DInput2, Dtow, EntropyAdd, EntropyBatchGmmTable,
                             EntropyContextNew, EntropyConv2Batch, EntropyCtxPadRun2, Extract, MultiProject,
                             PseudoContextV2, PseudoDQUANT, PseudoFillV2, PseudoQUANTV2, SphereSlice,
                             SphereUslice, SSIM, backend)
from .model_zoo_v2 import ClipData, DecoderV2, EncoderV2
from . import container

psnr_f = lambda xa: 10 * math.log10(1. / xa)

model_ssim_list = ['1_56', '2_56', '3_56', '4_56', '5_112', '6_112', '7_112', '8_192', '9_192']
ssim_channel_list = [56, 56, 56, 56, 112, 112, 112, 192, 192]
model_mse_list = ['1_56', '2_56', '3_56', '4_112', '5_112', '6_112', '7_112', '8_192', '9_192', '10_192']
mse_channel_list = [56, 56, 56, 112, 112, 112, 112, 192, 192, 192]
mse_model_dir = './demo/mse'
ssim_model_dir = './demo/ssim'

NPART = 16          # latitude tiles (pseudo_codec.py:166)
CHANNELS = 192      # transform width and code channels
QUANT_LEVELS = 8
DOWN = 16           # spatial down-sampling of the analysis transform


def latent_shape(height, width, npart=NPART):
    """(rows per tile, columns) of the code tensor for an ERP of height x width."""
    if height % (npart * DOWN) or width % DOWN:
        raise ValueError('ERP size %dx%d: height must be a multiple of %d and width of %d'
                         % (height, width, npart * DOWN, DOWN))
    return height // npart // DOWN, width // DOWN


class EntropyConvDBT(nn.Module):
    """causal halo update + masked conv with three weight sets (reference: pseudo_codec.py:27-38)."""

    def __init__(self, batch, ngroups, cin, cout, hidden, npart, out_layer, ctx, device_id, act=True):
        super(EntropyConvDBT, self).__init__()
        self.pad = EntropyCtxPadRun2(2, npart, ngroups, ctx, not hidden, device=device_id)
        self.conv = EntropyConv2Batch(npart, ngroups, cin, cout, 5, ctx, 2, 0 if out_layer else 2, batch=batch,
                                      hidden=hidden, act=act, device=device_id)

    def forward(self, x):
        return self.conv(self.pad(x))


class EntropyResidualBlockDBT(nn.Module):
    """(reference: pseudo_codec.py:40-51)"""

    def __init__(self, batch, ngroups, cpn, npart, ctx, device_id=0):
        super(EntropyResidualBlockDBT, self).__init__()
        self.conv1 = EntropyConvDBT(batch, ngroups, cpn, cpn, True, npart, False, ctx, device_id, True)
        self.conv2 = EntropyConvDBT(batch, ngroups, cpn, cpn, True, npart, False, ctx, device_id, True)
        self.add = EntropyAdd(npart, cpn * ngroups, ngroups, 2, ctx, device=device_id)

    def forward(self, x):
        return self.add(self.conv2(self.conv1(x)), x)


_STEPPED = (EntropyConv2Batch, EntropyCtxPadRun2, EntropyAdd, DInput2, DExtract2, DExtract2Batch)


@torch.no_grad()
def restart_entropy_network(m):
    """reset the step counter of every wavefront op (reference: pseudo_codec.py:53-66)"""
    if isinstance(m, _STEPPED):
        m.restart()


class _EntropyModel(nn.Module):
    """what the encoder and decoder sides share: context, input scatter, the
    12-layer three-headed masked network, GMM table"""

    def __init__(self, ngroup, npart, opt_f, bin_num, gid):
        super(_EntropyModel, self).__init__()
        self.cuda = backend.device_of(gid)
        self.ctx2 = EntropyContextNew(npart, opt=opt_f, device=gid)
        self.ipt = DInput2(ngroup, npart, self.ctx2, 2, -3.5, 3, device=gid)
        self.npart, self.ngroup = npart, ngroup
        self.fill = PseudoFillV2(0, npart, self.ctx2, 0, device=gid)
        self.mcoder = None
        self.bias = (bin_num - 1) / 2.
        layers = [EntropyConvDBT(3, ngroup, 1, 3, False, npart, False, self.ctx2, gid, True)]
        layers += [EntropyResidualBlockDBT(3, ngroup, 3, npart, self.ctx2, gid) for _ in range(5)]
        layers += [EntropyConvDBT(3, ngroup, 3, 3, True, npart, True, self.ctx2, gid, False)]
        self.net = nn.Sequential(*layers)
        self.ext = DExtract2Batch(npart, ngroup, self.ctx2, device=gid)
        self.gmm = EntropyBatchGmmTable(bin_num, self.bias, 3, 65536, device=gid)
        backend.watch_state_dict(self)  # reloaded weights drop the engine's repacked slabs

    def start(self, code_name='./tmp/data'):
        self.apply(restart_entropy_network)
        self.mcoder = backend.coder().coder(code_name)

    def steps(self, h_full, w):
        return h_full + w + self.ngroup - 2

    def tables(self, packed):
        """one wavefront step: scatter `packed` symbols of the previous step, run
        the network, return (integer CDF rows on the CPU, number of rows)"""
        b = self.ipt(packed)
        z, le = self.ext(self.net(b))
        vec = self.gmm(z, le)
        return b, vec.type(torch.int32).to('cpu'), int(le[0].item())


class EntEncoder(_EntropyModel):
    """(reference: pseudo_codec.py:68-114)"""

    def __init__(self, ngroup, npart=16, opt_f=True, bin_num=8, gid=0):
        super(EntEncoder, self).__init__(ngroup, npart, opt_f, bin_num, gid)
        self.ext_label = DExtract2(npart, ngroup, True, self.ctx2, device=gid)
        self.net = self.net.to(self.cuda)

    def forward(self, data):
        with torch.no_grad():
            data = self.fill(data)
            h, w = data.shape[2:]
            self.ctx2.setup_context(w)
            self.mcoder.start_encoder()
            h_full = h * self.npart
            label = torch.zeros((1, 1, h_full, w), dtype=torch.float32).to(self.cuda)
            for _ in range(self.steps(h_full, w)):
                _, pred, ln = self.tables(label)
                label, _ = self.ext_label(data)
                self.mcoder.encodes(pred, 8, label.type(torch.int32).to('cpu'), ln)
            self.mcoder.end_encoder()


class EntDecoder(_EntropyModel):
    """(reference: pseudo_codec.py:117-160)"""

    def __init__(self, ngroup, npart=16, opt_f=True, bin_num=8, gid=0):
        super(EntDecoder, self).__init__(ngroup, npart, opt_f, bin_num, gid)
        self.net = self.net.to(self.cuda)

    def forward(self, h, w):
        with torch.no_grad():
            self.ctx2.setup_context(w)
            self.mcoder.start_decoder()
            h_full = h * self.npart
            pout = torch.zeros((1, 1, h_full, w), dtype=torch.float32).to(self.cuda)
            b = None
            for _ in range(self.steps(h_full, w)):
                b, pred, ln = self.tables(pout)
                pout = self.mcoder.decodes(pred.view(-1, 9), 8, ln).to(self.cuda).view(1, 1, h_full, w).contiguous()
            code = (b[:self.npart, :, 2:-2, 2:-2] + self.bias).contiguous()
            return self.fill(code)


class PseudoEncoder(nn.Module):
    """ERP image -> code file (reference: pseudo_codec.py:162-186)"""

    def __init__(self, valid_dim, device_id):
        super(PseudoEncoder, self).__init__()
        npart, opt = NPART, True
        dev = backend.device_of(device_id)
        self.slice = SphereSlice(npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(npart, opt, device=device_id)
        self.encoder = EncoderV2(CHANNELS, CHANNELS, npart, self.ctx, device_id).to(dev)
        self.quant = PseudoQUANTV2(CHANNELS, QUANT_LEVELS, npart, self.ctx, device_id=device_id, ntop=2)
        self.valid_dim = valid_dim
        self.ext = Extract(valid_dim)
        self.mean_val = (QUANT_LEVELS - 1) / 2.
        self.dtw = Dtow(2, True, device_id)
        self.ent = EntEncoder(valid_dim // 4, npart, opt, QUANT_LEVELS, gid=device_id)
        backend.watch_state_dict(self)

    def symbols(self, x):
        """quantiser indices in wavefront layout (npart, valid_dim/4, 2h, 2w)"""
        with torch.no_grad():
            _, code_i = self.quant(self.encoder(self.slice(x)))
            return self.dtw(self.ext(code_i))

    def forward(self, x, code_name, header=None):
        """x -> code file.  On the GPU the entropy stage runs on the native engine
        (engine.EntropyEngine: one launch per layer over all wavefront steps); the file is
        byte for byte what the op-by-op loop of `forward_per_op` writes
        (tests/test_gpu_engine.py).  PCONV_ENTROPY=per-op forces the loop.
        header: dict(model_idx=, ssim=) -> the file gets the 16-byte container header
        (container.py) in front of the same payload; None = the reference's raw stream."""
        with torch.no_grad():
            hcode_i = self.symbols(x)
            eng = _native_engine(self, "enc", hcode_i)
            if eng is None:
                self.ent.start(code_name)
                self.ent(hcode_i)
            else:
                stream = eng.encode(self.ent.fill(hcode_i).contiguous())[0]
                with open(code_name, "wb") as f:
                    f.write(stream)
            if header is not None:
                with open(code_name, "rb") as f:
                    payload = f.read()
                container.write(code_name, payload, height=x.shape[2], width=x.shape[3],
                                model_idx=header["model_idx"], ssim=header["ssim"], valid_dim=self.valid_dim)

    def forward_per_op(self, x, code_name):
        """the reference's loop (pseudo_codec.py:97-114): ~36 op calls per wavefront step"""
        with torch.no_grad():
            hcode_i = self.symbols(x)
            self.ent.start(code_name)
            self.ent(hcode_i)


class PseudoDecoder(nn.Module):
    """code file -> ERP image (reference: pseudo_codec.py:188-213).  height/width
    default to the reference's only size."""

    def __init__(self, valid_dim, device_id):
        super(PseudoDecoder, self).__init__()
        self.npart, opt, self.channels, self.code_channels = NPART, True, CHANNELS, CHANNELS
        dev = backend.device_of(device_id)
        self.valid_dim = valid_dim
        self.uslice = SphereUslice(self.npart, pad=0, opt=opt, device=device_id)
        self.ctx = PseudoContextV2(self.npart, opt, device=device_id)
        self.decoder = DecoderV2(self.channels, self.code_channels, self.npart, self.ctx, device_id).to(dev)
        self.clip = ClipData()
        self.quant = PseudoDQUANT(self.code_channels, QUANT_LEVELS, self.npart, self.ctx, device_id=device_id)
        self.wtd = Dtow(2, False, device_id)
        self.ent = EntDecoder(self.valid_dim // 4, self.npart, opt, QUANT_LEVELS, gid=device_id)
        backend.watch_state_dict(self)

    def reconstruct(self, hcode_i):
        """symbols in wavefront layout -> image"""
        with torch.no_grad():
            code_ext = self.quant(self.wtd(hcode_i))
            code_f = torch.zeros((code_ext.shape[0], self.code_channels) + tuple(code_ext.shape[2:])).type_as(code_ext)
            code_f[:, :self.valid_dim] = code_ext
            return self.clip(self.uslice(self.decoder(code_f.contiguous())))

    def forward(self, code_name, height=512, width=1024, raw=None):
        """code file -> image; the entropy stage on the native engine when on the GPU
        (see PseudoEncoder.forward).  raw=True: the reference's headerless stream, size from the
        arguments; raw=False: the file carries the container header (container.py) and height /
        width are read from it; raw=None (default): a file that starts with a valid container
        header is read as one, anything else as a raw stream."""
        with torch.no_grad():
            if raw is None:
                raw = container.sniff(code_name) is None
            if not raw:
                head, payload = container.read(code_name)
                if head["valid_dim"] != self.valid_dim:
                    raise container.ContainerError("file was coded with valid_dim %d, this decoder has %d"
                                                   % (head["valid_dim"], self.valid_dim))
                height, width = head["height"], head["width"]
            h, w = latent_shape(height, width, self.npart)
            eng = _native_engine(self, "dec", None, 2 * h, 2 * w)
            if eng is None:
                if not raw:   # the coder modules read files: hand them the bare payload
                    code_name = code_name + ".payload"
                    with open(code_name, "wb") as f:
                        f.write(payload)
                self.ent.start(code_name)
                try:
                    return self.reconstruct(self.ent(2 * h, 2 * w))
                finally:
                    if not raw:
                        self.ent.mcoder = None
                        os.remove(code_name)
            if raw:
                with open(code_name, "rb") as f:
                    payload = f.read()
            return self.reconstruct(eng.decode([payload]))

    def forward_per_op(self, code_name, height=512, width=1024):
        """the reference's loop (pseudo_codec.py:145-160)"""
        with torch.no_grad():
            h, w = latent_shape(height, width, self.npart)
            self.ent.start(code_name)
            return self.reconstruct(self.ent(2 * h, 2 * w))


def _native_engine(codec, which, symbols=None, h2=None, w2=None):
    """single-frame EntropyEngine bound to codec.ent, cached on the codec per latent size;
    None when the active backend has no native engine (the CPU oracle used by the tests)
    or PCONV_ENTROPY=per-op asks for the op-by-op loops"""
    import os
    ops = backend.ops()
    if os.environ.get("PCONV_ENTROPY", "native") == "per-op" or not getattr(ops, "FUSED_EPILOGUE", False):
        return None
    if symbols is not None:
        if not symbols.is_cuda:
            return None
        h2, w2 = symbols.shape[2], symbols.shape[3]
    from .engine import EntropyEngine
    cache = codec.__dict__.setdefault("_pconv_engines", {})
    key = (which, int(h2), int(w2))
    if key not in cache:
        dev = next(codec.ent.parameters()).device
        if dev.type != "cuda":
            return None
        cache[key] = EntropyEngine(codec.ent, h2, w2, 1, dev)
    else:
        cache[key].bind(codec.ent)  # parameters may have been reloaded
    return cache[key]


# -- image / checkpoint I/O ---------------------------------------------------
def read_image(path):
    """uint8 HxWx3 in BGR order (what cv2.imread returns in the reference)"""
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1].copy()


def write_image(path, img_bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1])).save(path)


def img2tensor(img, device):
    ts = torch.from_numpy(img.transpose(2, 0, 1).astype(np.float32)) / 255.
    return torch.unsqueeze(ts, 0).to(device).contiguous()


def tensor2img(data):
    return (data[0] * 255.).to('cpu').detach().numpy().transpose(1, 2, 0).astype(np.uint8)


def check_img(img, height=512, width=1024):
    """bicubic resize to the coding size when the input differs (reference: pseudo_codec.py:229-234)"""
    if img.shape[0] == height and img.shape[1] == width:
        return img
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize((width, height), Image.BICUBIC))


def load_models(model, p1, p2, device):
    merged = OrderedDict(**torch.load(p1, map_location=device), **torch.load(p2, map_location=device))
    model.load_state_dict(merged)


def _pick(model_idx, mse):
    prex = model_mse_list[model_idx] if mse else model_ssim_list[model_idx]
    vd = mse_channel_list[model_idx] if mse else ssim_channel_list[model_idx]
    return prex, vd, (mse_model_dir if mse else ssim_model_dir)


def bitrate(path, height=512, width=1024):
    """bits per pixel of the coded payload (a container header is not counted, so the figure is
    the reference's `os.path.getsize(fc)*8/1024./512.` for the same stream, pseudo_codec.py:247,283)"""
    head = container.sniff(path)
    nbytes = os.path.getsize(path) - (container.HEADER_BYTES if head is not None else 0)
    return nbytes * 8 / float(width) / float(height)


def encoding(img_list, out_list, model_idx=0, mse=True, device_id=0, height=512, width=1024, boxed=False):
    """reference: pseudo_codec.py:236-247.  The files are the reference's headerless streams unless
    boxed=True (--container): then the 16-byte header of container.py goes in front of the same
    payload and the file decodes without any size / model argument."""
    prex, vd, model_dir = _pick(model_idx, mse)
    dev = backend.device_of(device_id)
    t1 = PseudoEncoder(vd, device_id=device_id).to(dev)
    load_models(t1, '{}/{}_encoder.pt'.format(model_dir, prex), '{}/{}_ent.pt'.format(model_dir, prex), dev)
    header = {"model_idx": model_idx, "ssim": not mse} if boxed else None
    for fn, fo in zip(img_list, out_list):
        data = img2tensor(check_img(read_image(fn), height, width), dev)
        t1(data, fo, header)
        print('Encoding {}, bitrate: {:.3f}bpp'.format(fn, bitrate(fo, height, width)))


def _decoder_for(code_list, model_idx, mse, device_id, raw):
    """decoder with its checkpoint loaded.  When the FIRST file carries a container header the
    model is the one it names (all container files of a call must agree); raw=True never looks"""
    head = None if raw else container.sniff(code_list[0])
    if head is not None:
        model_idx, mse = head["model_idx"], not head["ssim"]
    prex, vd, model_dir = _pick(model_idx, mse)
    dev = backend.device_of(device_id)
    t1 = PseudoDecoder(vd, device_id=device_id).to(dev)
    load_models(t1, '{}/{}_decoder.pt'.format(model_dir, prex), '{}/{}_ent.pt'.format(model_dir, prex), dev)
    return t1, dev, model_idx, mse


def _file_geometry(fc, model_idx, mse, height, width, raw):
    """(height, width, is_raw) of one code file: from its container header when it has one (which
    must name the model the decoder was built for), else from the arguments"""
    head = None if raw else container.sniff(fc)
    if head is None:
        return height, width, True
    if head["model_idx"] != model_idx or head["ssim"] == mse:
        raise container.ContainerError("%s was coded with another model than the first file of the list" % fc)
    return head["height"], head["width"], False


def decoding(code_list, decoded_img_list, model_idx=0, mse=True, device_id=0, height=512, width=1024, raw=False):
    """reference: pseudo_codec.py:249-260"""
    t1, dev, model_idx, mse = _decoder_for(code_list, model_idx, mse, device_id, raw)
    for fc, fo in zip(code_list, decoded_img_list):
        h, w, is_raw = _file_geometry(fc, model_idx, mse, height, width, raw)
        write_image(fo, tensor2img(t1(fc, h, w, is_raw)))
        print('Decoding {}, output to {}'.format(fc, fo))


class ViewportMetrics(object):
    """viewport PSNR / SSIM of the paper: 14 rectilinear views, 171x256, FoV pi/2
    (reference: pseudo_codec.py:270-282)"""

    def __init__(self, device_id=0):
        dev = backend.device_of(device_id)
        self.pr1 = MultiProject(171, int(171 * 1.5), 0.5, False, device_id).to(dev)
        self.pr2 = MultiProject(171, int(171 * 1.5), 0.5, False, device_id).to(dev)
        self.sim_func = SSIM(11, 3).to(dev)

    def __call__(self, original, decoded):
        x, y = self.pr1(original), self.pr2(decoded)
        mse_loss = torch.mean((x - y) ** 2).item()
        return psnr_f(mse_loss), self.sim_func(x, y).item()


def decoding_and_test(code_list, img_list, model_idx=0, mse=True, device_id=0, height=512, width=1024, raw=False):
    """reference: pseudo_codec.py:263-290"""
    t1, dev, model_idx, mse = _decoder_for(code_list, model_idx, mse, device_id, raw)
    metrics = ViewportMetrics(device_id)
    rows = []
    for fc, fn in zip(code_list, img_list):
        h, w, is_raw = _file_geometry(fc, model_idx, mse, height, width, raw)
        rdata = t1(fc, h, w, is_raw)
        data = img2tensor(check_img(read_image(fn), h, w), dev)
        pr, vssim = metrics(data, rdata)
        rt = bitrate(fc, h, w)
        rows.append((rt, pr, vssim))
        print('Decoding {}, compare it to {} \n Bitrate:{:.3f}bpp, PSNR:{:.2f}dB, SSIM:{:.4f}'.format(fc, fn, rt, pr, vssim))
    print('-' * 53 + '\nAverage Performance\n' + '-' * 53)
    rt, pr, vssim = np.average(np.array(rows), axis=0)
    print('Bitrate:{:.3f}bpp, PSNR:{:.2f}dB, SSIM:{:.4f}'.format(rt, pr, vssim))
    return rows


def read_list(fname):
    with open(fname) as f:
        return [line.rstrip('\n') for line in f.readlines()]


def check_models():
    assert os.path.exists('{}/{}_encoder.pt'.format(mse_model_dir, model_mse_list[0])), \
        'Please make sure the pretrained models for VMSE exists in the mse_model_dir'
    assert os.path.exists('{}/{}_encoder.pt'.format(ssim_model_dir, model_ssim_list[0])), \
        'Please make sure the pretrained models for VSSIM exists in the ssim_model_dir'


def main(argv=None):
    parser = argparse.ArgumentParser(description='Pseudo Convolution for 360 Image Compression')
    parser.add_argument('--img-list', nargs='*', help='The image list contains the input images for encoding and testing')
    parser.add_argument('--code-list', nargs='*', help='The code file list for codes')
    parser.add_argument('--out-list', nargs='*', help='The out list for saving decoded images.')
    parser.add_argument('--img-file', help='The file contains the input images for encoding and testing')
    parser.add_argument('--code-file', help='The file contains the list for codes')
    parser.add_argument('--out-file', help='The file  contains the names of decoded images.')
    parser.add_argument('--model-idx', type=int, default=0, help='Model index (0-9) for VMSE, (0-8) for VSSIM')
    parser.add_argument('--enc', action='store_true', default=False, help='Encoding flag, set for encoding phase.')
    parser.add_argument('--dec', action='store_true', default=False, help='Decoding flag, set for decoding phase.')
    parser.add_argument('--test', action='store_true', default=False, help='Testing flag, set for decoding and evalating the performance.')
    parser.add_argument('--ssim', action='store_true', default=False,
                        help='Default with models optimized for VMSE, set this flag for choosing the models optimized for VSSIM')
    parser.add_argument('--gpu-id', type=int, default=0, help='The graphic card id for encoding and decoding.')
    parser.add_argument('--height', type=int, default=512, help='ERP height of the coded image (multiple of 256)')
    parser.add_argument('--width', type=int, default=1024, help='ERP width of the coded image (multiple of 16)')
    parser.add_argument('--container', action='store_true', default=False,
                        help='Encoding: put a 16-byte header (size, model, valid_dim, length) in front of the stream, so '
                             "that the file decodes without --height/--width/--model-idx/--ssim.  Default: the reference's "
                             'headerless files.  Decoding recognises such files by their magic')
    parser.add_argument('--raw', action='store_true', default=False,
                        help='Decoding: never look for a container header (size and model from the flags)')
    args = parser.parse_args(argv)
    check_models()
    midx = args.model_idx
    if args.ssim:
        assert 0 <= midx < 9, '(0-8) for VSSIM'
    else:
        assert 0 <= midx < 10, '(0-9) for VMSE'
    assert args.enc or args.dec or args.test, \
        'Should set one flag, (--enc) for encoding, (--dec) for decoding, (--test) for testing.'
    pick = lambda lst, fil: lst if lst is not None else (read_list(fil) if fil is not None else None)
    img_list, code_list, out_list = pick(args.img_list, args.img_file), pick(args.code_list, args.code_file), \
        pick(args.out_list, args.out_file)
    size = dict(height=args.height, width=args.width)
    if args.enc:
        assert img_list is not None, 'No input images for encoding'
        assert code_list is not None, 'No code files for saving the codes'
        assert len(img_list) == len(code_list), 'The number of images and codes should be the same'
        encoding(img_list, code_list, midx, not args.ssim, args.gpu_id, boxed=args.container and not args.raw, **size)
    else:
        assert code_list is not None, 'No code files for decoding'
        if args.dec:
            assert out_list is not None, 'No out files for saving the decoded images'
            assert len(code_list) == len(out_list), 'The number of codes and reconstructed images should be the same'
            decoding(code_list, out_list, midx, not args.ssim, args.gpu_id, raw=args.raw, **size)
        else:
            assert img_list is not None, 'No source images for evaluation.'
            assert len(code_list) == len(img_list), 'The number of codes and corresponding source images should be the same'
            decoding_and_test(code_list, img_list, midx, not args.ssim, args.gpu_id, raw=args.raw, **size)


if __name__ == '__main__':
    main()
