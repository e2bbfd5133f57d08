"""Batched codec on the native entropy engine.

`EntropyEngine` binds the C++ wavefront loop of libpconv_hip.so
(include/pconv_hip.h, pconv_ee_*) to the parameters of an EntEncoder / EntDecoder
module.  `CodecEngine` runs whole frames: transforms through the operator layer
(hand-written tile convolution, pad, fill ... kernels), entropy coding through the
engine, several frames in lock-step.  The streams are byte-identical to the ones
the per-op PseudoEncoder path writes, so either side can decode the other's files.
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _native
from ._native import PconvError, call
from .PCONV_operator import backend, set_weight
from . import pseudo_codec as PC


class EntropyEngine(object):
    """nimg frames in lock-step through the 12-layer context model
    (reference loops: pseudo_codec.py:97-114, 145-160)."""

    def __init__(self, ent, h, w, nimg, device):
        self.lib = _native.hip_lib()
        self.device = torch.device(device)
        self.h, self.w, self.nimg = int(h), int(w), int(nimg)
        self.npart, self.ngroup = ent.npart, ent.ngroup
        weight = np.asarray(set_weight(self.npart, True), dtype=np.float32)
        with torch.cuda.device(self.device):
            self.handle = self.lib.pconv_ee_create(self.npart, self.ngroup, self.h, self.w, self.nimg,
                                                   weight.ctypes.data, float(ent.bias), 8, 65536.0, 1e-6)
        if not self.handle:
            raise PconvError("pconv_ee_create failed: %s" % (self.lib.pconv_last_error() or b"").decode())
        self.bind(ent)

    def bind(self, ent):
        """point the engine at the module's parameters (kept alive here)"""
        convs = [ent.net[0].conv]
        for b in range(1, 6):
            convs += [ent.net[b].conv1.conv, ent.net[b].conv2.conv]
        convs.append(ent.net[6].conv)
        signature = (backend.param_epoch(),) + tuple(
            (t.data_ptr(), t._version) for conv in convs
            for t in (conv.weight, conv.bias, conv.relu if conv.act else None) if t is not None)
        if getattr(self, "_signature", None) == signature:
            return  # same tensors, unmodified since the last bind
        self._signature = signature
        self._params = []
        for layer, conv in enumerate(convs):
            tensors = [conv.weight, conv.bias, conv.relu if conv.act else None]
            held = []
            for t in tensors:
                if t is None:
                    held.append(None)
                    continue
                t = t.detach()
                if t.device != self.device or not t.is_contiguous() or t.dtype != torch.float32:
                    t = t.to(self.device, torch.float32).contiguous()
                held.append(t)
            self._params.append(held)
            with torch.cuda.device(self.device):
                call("pconv_ee_set_layer", self.handle, layer, held[0].data_ptr(), held[1].data_ptr(),
                     held[2].data_ptr() if held[2] is not None else None,
                     torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.pconv_ee_destroy(h)

    @property
    def symbols_per_image(self):
        return int(self.lib.pconv_ee_symbols_per_image(self.handle))

    def _check_symbols(self, symbols):
        expect = (self.nimg * self.npart, self.ngroup, self.h, self.w)
        if tuple(symbols.shape) != expect or not symbols.is_cuda or not symbols.is_contiguous():
            raise PconvError("EntropyEngine.encode: expected contiguous GPU tensor %s, got %s" % (expect, tuple(symbols.shape)))

    def _streams(self):
        out = []
        for i in range(self.nimg):
            n = ctypes.c_size_t(0)
            p = self.lib.pconv_ee_stream(self.handle, i, ctypes.byref(n))
            out.append(ctypes.string_at(p, n.value))
        return out

    def encode(self, symbols):
        """symbols (nimg*npart, ngroup, h, w) float indices, dead columns zero -> [bytes] per frame"""
        self._check_symbols(symbols)
        call("pconv_ee_set_encode_ranges", self.handle, 0)  # (the default: a lone call's coding is all tail)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            call("pconv_ee_encode", self.handle, symbols.data_ptr(), stream)
        return self._streams()

    def encode_begin(self, symbols, ranges=0):
        """first half of encode: queues the GPU part and starts the host coder thread, returns at
        once; the caller may queue other GPU work before encode_end().  ranges: step ranges the call's
        last group is evaluated in (0 = the engine's default, 1 = one piece: for calls whose coding
        hides under later GPU work)"""
        self._check_symbols(symbols)
        call("pconv_ee_set_encode_ranges", self.handle, int(ranges))
        self._pending = symbols  # kept alive until encode_end
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            call("pconv_ee_encode_begin", self.handle, symbols.data_ptr(), stream)

    def encode_end(self):
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            try:
                call("pconv_ee_encode_end", self.handle, stream)
            finally:
                self._pending = None
        return self._streams()

    def decode(self, streams):
        if len(streams) != self.nimg:
            raise PconvError("EntropyEngine.decode: %d streams for %d frames" % (len(streams), self.nimg))
        bufs = [ctypes.create_string_buffer(s, len(s)) for s in streams]
        ptrs = (ctypes.c_void_p * self.nimg)(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
        sizes = (ctypes.c_size_t * self.nimg)(*[len(s) for s in streams])
        out = torch.empty((self.nimg * self.npart, self.ngroup, self.h, self.w), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            call("pconv_ee_decode", self.handle, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p),
                 out.data_ptr(), stream)
        return out


class CodecEngine(object):
    """frames in, byte streams out, and back: PseudoEncoder / PseudoDecoder
    (pseudo_codec.py:162-213) with the entropy loops on the native engine."""

    def __init__(self, valid_dim=56, device_id=0, encoder=None, decoder=None):
        self.device_id = device_id
        self.device = torch.device("cuda", device_id)
        self.enc = encoder if encoder is not None else PC.PseudoEncoder(valid_dim, device_id)
        self.dec = decoder if decoder is not None else PC.PseudoDecoder(valid_dim, device_id)
        self._engines = {}
        # bench.py: a list that receives (phase, start event, end event) per call -- analysis, entropy_encode,
        # entropy_decode, synthesis -- recorded in the caller's stream (None: no events)
        self.phase_probe = None

    def _mark(self):
        if self.phase_probe is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def _phase(self, name, e0, e1):
        if self.phase_probe is not None and e0 is not None and e1 is not None:
            self.phase_probe.append((name, e0, e1))

    def _engine(self, which, h, w, nimg, slot=0):
        key = (which, h, w, nimg, slot)
        ent = self.enc.ent if which == "enc" else self.dec.ent
        if key not in self._engines:
            self._engines[key] = EntropyEngine(ent, h, w, nimg, self.device)
        else:
            # the engine owns repacked copies of the weights: re-bind on every fetch
            # (a no-op while the parameters are unchanged) so that a reloaded or edited
            # entropy model can never run on stale slabs
            self._engines[key].bind(ent)
        return self._engines[key]

    # Transform stages from 1/4 scale down do not fill the chip on one frame (a 3x3 192->192 launch at
    # 1/8 scale has 16 x fewer workgroups than at 1/2 scale; at 1/4 scale the last of its 6.5 rounds of
    # workgroups is half empty): blocks [0, ANALYSIS_SPLIT) of EncoderV2.net run frame by frame (their
    # activations are GBs at 4096x2048), the rest -- 1/4 scale and below -- on all frames of the call at
    # once; likewise the first SYNTHESIS_SPLIT blocks of DecoderV2.net.  A tile's outputs do not depend
    # on how many tiles a launch carries (same kernel, same per-tile arithmetic): bit-identical either
    # way (tests/test_codec_cpu.py on the oracle; on the GPU at the metric size:
    # tests/test_gpu_codec_vs_oracle.py::test_benchmarked_workload_eight_frames_at_the_metric_size).  0 = whole transform frame by frame.  Measured (8 frames,
    # profiles/round3_split_transforms.txt): 0/0 52.6, 6/5 54.0, 3/8 54.5 MPix/s.
    ANALYSIS_SPLIT = int(os.environ.get("PCONV_ANALYSIS_SPLIT", "3"))
    SYNTHESIS_SPLIT = int(os.environ.get("PCONV_SYNTHESIS_SPLIT", "8"))

    @staticmethod
    def _split(knob, net, tail):
        """the knob clamped to a split point that leaves the fused tail of the transform (the final
        conv + sigmoid of EncoderV2: 1 module; the final conv + depth-to-width of DecoderV2: 2) on the
        batched / per-frame side it belongs to; out-of-range values mean 'no split'"""
        k = int(knob)
        return k if 0 < k <= len(net) - tail else 0

    @staticmethod
    def _frame_of(x, i, tiles):
        """tile-batch rows of frame i of a batched activation; keeps the padded-buffer tag so that
        the consumer's PseudoPad still only fills the ring"""
        y = x[i * tiles:(i + 1) * tiles]
        ring = getattr(x, "_pconv_ring", None)
        if ring is not None:
            y._pconv_ring = (ring[0][i * tiles:(i + 1) * tiles], ring[1])
        return y

    @staticmethod
    def _batch_buffer(like, n):
        """(n * tiles, C, h, w) tensor for n frames' worth of `like`; when `like` lives inside a padded
        buffer (its producer wrote it there for the consumer's PseudoPad) so does the batch, so that the
        first batched block still only fills the ring instead of copying the tensor"""
        tn, c, h, w = like.shape
        ring = getattr(like, "_pconv_ring", None)
        if ring is None:
            return torch.empty((n * tn, c, h, w), dtype=like.dtype, device=like.device)
        r = ring[1]
        buf = torch.empty((n * tn, c, h + 2 * r, w + 2 * r), dtype=like.dtype, device=like.device)
        out = buf[:, :, r:-r, r:-r]
        out._pconv_ring = (buf, r)
        return out

    @torch.no_grad()
    def symbols(self, frames):
        """(n, 3, H, W) -> quantiser indices (n*16, valid_dim/4, 2h, 2w), dead columns zero."""
        enc, n = self.enc, frames.shape[0]
        k = self._split(self.ANALYSIS_SPLIT, enc.encoder.net, 1) if hasattr(enc.encoder, "forward_range") else 0
        if n == 1 or k <= 0:
            per_frame = [enc.ent.fill(enc.symbols(frames[i:i + 1])).clone() for i in range(n)]
            return per_frame[0] if len(per_frame) == 1 else torch.cat(per_frame, 0)
        mid = None
        # all resident frames through SphereSlice in ONE launch (1.6 GB instead of 8 x 0.2 GB: the frame-wise
        # launches were launch-size-bound at 2.4 TB/s); the op owns the tile stack until its next call
        tiles = enc.ent.npart
        stack = enc.slice(frames.contiguous())
        for i in range(n):
            m = enc.encoder.forward_range(stack[i * tiles:(i + 1) * tiles], 0, k)
            if mid is None:
                mid = self._batch_buffer(m, n)
            mid[i * m.shape[0]:(i + 1) * m.shape[0]].copy_(m)
            del m
        code = enc.encoder.forward_range(mid, k, len(enc.encoder.net))
        del mid
        _, code_i = enc.quant(code)
        return enc.ent.fill(enc.dtw(enc.ext(code_i))).clone()

    @torch.no_grad()
    def reconstruct(self, sym, n):
        """decoded symbols of n frames (n*16, valid_dim/4, 2h, 2w) -> (n, 3, H, W)"""
        dec, tiles = self.dec, self.dec.npart
        k = self._split(self.SYNTHESIS_SPLIT, dec.decoder.net, 2) if hasattr(dec.decoder, "forward_range") else 0
        if n == 1 or k <= 0:
            out = [dec.reconstruct(sym[i * tiles:(i + 1) * tiles]).clone() for i in range(n)]
            return out[0] if n == 1 else torch.cat(out, 0)
        code_ext = dec.quant(dec.wtd(sym))
        code_f = torch.zeros((code_ext.shape[0], dec.code_channels) + tuple(code_ext.shape[2:]), dtype=code_ext.dtype,
                             device=code_ext.device)
        code_f[:, :dec.valid_dim] = code_ext
        mid = dec.decoder.forward_range(code_f.contiguous(), 0, k)
        ops = backend.ops()
        usl = dec.uslice.native(mid)
        direct = hasattr(usl, "forward_into") and hasattr(ops, "leaky_clip_")
        out, batch = [], None
        for i in range(n):
            tx = dec.decoder.forward_range(self._frame_of(mid, i, tiles), k, len(dec.decoder.net))
            if direct:
                # SphereUslice writes frame i of the result in place and ClipData runs once, in place, over all
                # frames (before: uslice into the op's buffer, clip into a new tensor, clone, and a concatenation
                # of the eight clones -- three more passes over 0.8 GB)
                if batch is None:
                    batch = torch.empty((n, tx.shape[1], tx.shape[2] * tiles, tx.shape[3]), dtype=tx.dtype, device=tx.device)
                usl.forward_into(tx.contiguous(), batch[i:i + 1])
            else:
                out.append(dec.clip(dec.uslice(tx)).clone())
        return ops.leaky_clip_(batch) if direct else torch.cat(out, 0)

    # frames entropy-coded per pipeline stage of encode(): the stage's tables are arithmetic-coded
    # on the CPU while the GPU runs the analysis transform of the following frames
    ENCODE_CHUNK = int(os.environ.get("PCONV_ENCODE_CHUNK", "2"))

    @torch.no_grad()
    def encode(self, frames):
        """(n, 3, H, W) frames on the GPU -> n byte strings.  The frames of a chunk go through the
        entropy wavefront in lock-step; the chunks are pipelined: chunk k's CDF tables are coded by
        host threads while the GPU computes the symbols of chunk k+1 (each chunk has its own
        engine: the coder reads the engine's pinned buffers until encode_end)."""
        n = frames.shape[0]
        chunk = self.ENCODE_CHUNK if n > self.ENCODE_CHUNK else n
        tiles = self.enc.ent.npart
        # with the split transform the symbols of ALL frames come out of one batched tail: the chunks
        # then only pipeline the entropy stage (GPU tables of chunk k+1 beside the CPU coding of chunk k)
        batched = n > 1 and hasattr(self.enc.encoder, "forward_range") and \
            self._split(self.ANALYSIS_SPLIT, self.enc.encoder.net, 1) > 0
        t0 = self._mark()
        sym_all = self.symbols(frames).contiguous() if batched else None
        t1 = self._mark()
        pending, out = [], []
        try:
            for k, lo in enumerate(range(0, n, chunk)):
                if batched:
                    sym = sym_all[lo * tiles:(lo + chunk) * tiles]
                else:
                    sym = self.symbols(frames[lo:lo + chunk]).contiguous()
                eng = self._engine("enc", sym.shape[2], sym.shape[3], sym.shape[0] // tiles, slot=k)
                # only the last chunk's arithmetic coding is a tail nothing hides: only it is worth step ranges
                eng.encode_begin(sym, ranges=0 if lo + chunk >= n else 1)
                pending.append(eng)
        except BaseException:
            # a later chunk failed: join the coder threads of the chunks already started, or their
            # engines would refuse every later encode ("the previous encode has not been ended")
            for eng in pending:
                try:
                    eng.encode_end()
                except Exception:
                    pass
            raise
        for eng in pending:
            out += eng.encode_end()
        if batched:
            # (every coder thread has been joined: the GPU part of the entropy stage is over, the end event marks now)
            self._phase("analysis", t0, t1)
            self._phase("entropy_encode", t1, self._mark())
        return out

    # frames per pipeline stage of decode(); 0 = decode all frames of a call together, then run the
    # synthesis transforms (PCONV_DECODE_CHUNK overrides)
    DECODE_CHUNK = int(os.environ.get("PCONV_DECODE_CHUNK", "0"))

    @torch.no_grad()
    def decode(self, streams, height, width):
        """n byte strings -> (n, 3, H, W).  With DECODE_CHUNK = c > 0 and more than c frames the call
        is pipelined: while the synthesis transform of chunk k runs, the entropy decoder of chunk k+1
        (a latency chain that leaves most of the GPU idle) runs beside it on a second stream, driven
        by a host thread; two engines alternate."""
        h, w = PC.latent_shape(height, width, self.dec.npart)
        n = len(streams)
        tiles = self.dec.npart
        chunk = self.DECODE_CHUNK
        if chunk <= 0 or n <= chunk:
            t0 = self._mark()
            sym = self._engine("dec", 2 * h, 2 * w, n).decode(streams)
            t1 = self._mark()
            rec = self.reconstruct(sym, n)
            self._phase("entropy_decode", t0, t1)
            self._phase("synthesis", t1, self._mark())
            return rec
        chunks = [streams[i:i + chunk] for i in range(0, n, chunk)]
        side = torch.cuda.Stream(device=self.device)
        box = {}

        def run(k):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(side):
                    box[k] = self._engine("dec", 2 * h, 2 * w, len(chunks[k]), slot=k % 2).decode(chunks[k])
            except BaseException as exc:  # re-raised by the caller's thread
                box[k] = exc

        run(0)
        out = []
        for k in range(len(chunks)):
            sym = box.pop(k)
            if isinstance(sym, BaseException):
                raise sym
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(side)
            sym.record_stream(main)  # allocated on `side`, read by the synthesis transform on `main`
            worker = None
            if k + 1 < len(chunks):
                worker = threading.Thread(target=run, args=(k + 1,))
                worker.start()
            out.append(self.reconstruct(sym, len(chunks[k])))
            if worker is not None:
                worker.join()
        return torch.cat(out, 0)


class FramePipe(object):
    """Frames between pinned host memory and HBM beside the codec's compute (the two ends of the reference's
    flow, pseudo_codec.py:236-247 and 249-268: imread -> img2tensor -> .cuda() ... tensor2img -> imwrite).

    The bus carries the uint8 image as the image file holds it (H, W, 3) -- a quarter of the reference's fp32
    bytes -- on two copy streams of its own, double-buffered: while the codec works on batch k, batch k + 1 is
    uploaded and batch k - 1's reconstruction is downloaded.  img2tensor's division and tensor2img's cast run on
    the device with the reference's arithmetic (PCONV.frames_u8_to_f32 / frames_f32_to_u8).  Copies are issued
    frame by frame (25 MB at 4096x2048: half a millisecond each) so that the entropy engine's small, latency-bound
    transfers never queue behind one long DMA."""

    def __init__(self, n, height, width, device):
        self.n, self.h, self.w = int(n), int(height), int(width)
        self.device = torch.device(device)
        self.ops = backend.ops()
        if not hasattr(self.ops, "frames_u8_to_f32"):
            raise PconvError("FramePipe needs the HIP backend")
        shape = (self.n, self.h, self.w, 3)
        # one copy stream per direction, created through the C ABI (pconv_stream_create) rather than taken from
        # torch's pool (whose first use creates 64 streams in the process).  NOTE for processes that SHARE one GPU
        # (bench.py --share-gpu, never a deployment): the HIP runtime multiplexes a process's streams onto 4 hardware
        # queues by default, and with these two streams more the engine's queued decoder chains of two such processes
        # ran 2.3 x slower (2 ranks x 4 frames: 70 instead of 94 MPix/s) until GPU_MAX_HW_QUEUES=8 was exported; one
        # process per GPU is indifferent to the knob at 1 / 2 / 4 / 8 frames per call (profiles/round6_rehearsal.txt)
        self._raw = []
        self.up, self.down = self._stream(), self._stream()
        self.dev_in = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.dev_out = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.host_out = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.frames = torch.empty((self.n, 3, self.h, self.w), dtype=torch.float32, device=self.device)
        ev = lambda: [torch.cuda.Event() for _ in range(2)]
        self.up_done, self.in_free, self.out_ready, self.down_done = ev(), ev(), ev(), ev()
        self._in_used, self._out_used = [False, False], [False, False]

    def _stream(self):
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            call("pconv_stream_create", ctypes.byref(handle))
        self._raw.append(handle)   # (lives as long as the process: a pipe is created once per workload)
        return torch.cuda.ExternalStream(handle.value, device=self.device)

    def prefetch(self, host_u8, slot):
        """queue the upload of a batch (pinned uint8 (n, H, W, 3)) into input slot `slot` on the upload stream"""
        if tuple(host_u8.shape) != tuple(self.dev_in[slot].shape) or host_u8.dtype != torch.uint8:
            raise PconvError("FramePipe.prefetch: uint8 %s expected" % (tuple(self.dev_in[slot].shape),))
        with torch.cuda.stream(self.up):
            if self._in_used[slot]:
                self.up.wait_event(self.in_free[slot])     # the conversion that read this slot last has run
            for i in range(self.n):
                self.dev_in[slot][i].copy_(host_u8[i], non_blocking=True)
            self.up_done[slot].record(self.up)

    def take(self, slot):
        """the uploaded batch of `slot` as float32 (n, 3, H, W) in the caller's stream (img2tensor's arithmetic)"""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.up_done[slot])
        self.ops.frames_u8_to_f32(self.dev_in[slot], self.frames)
        self.in_free[slot].record(cur)
        self._in_used[slot] = True
        return self.frames

    def give(self, rec, slot):
        """queue tensor2img + the download of a reconstruction (n, 3, H, W) into host_out[slot]; returns that
        pinned tensor (valid once down_done[slot] has completed: wait(slot))"""
        cur = torch.cuda.current_stream(self.device)
        if self._out_used[slot]:
            cur.wait_event(self.down_done[slot])           # the previous download out of this slot has finished
        self.ops.frames_f32_to_u8(rec.contiguous(), self.dev_out[slot])
        self.out_ready[slot].record(cur)
        with torch.cuda.stream(self.down):
            self.down.wait_event(self.out_ready[slot])
            for i in range(self.n):
                self.host_out[slot][i].copy_(self.dev_out[slot][i], non_blocking=True)
            self.down_done[slot].record(self.down)
        self._out_used[slot] = True
        return self.host_out[slot]

    def wait(self, slot):
        self.down_done[slot].synchronize()
        return self.host_out[slot]
