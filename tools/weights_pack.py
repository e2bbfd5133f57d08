#!/usr/bin/env python3
"""Checkpoints that fit gpurun's 64 MiB return channel.

The codec has 17 M parameters (68 MB as fp32).  A GPU call can hand back 64 MiB under gpurun_out/, so
a training run returns its weights PACKED: tensors of >= 65536 elements (the convolution weights) as
fp16, everything else (biases, PReLU slopes, GDN beta / gamma, quantiser levels and counts) as fp32.
`unpack` restores fp32 state dicts -- the fp16-rounded weights ARE the trained model every later
measurement uses (the rounding happens once, before any evaluation).

    python tools/weights_pack.py pack   <out.pack.pt> name=<state.pt> [name=<state.pt> ...]
    python tools/weights_pack.py unpack <in.pack.pt> <out_dir>        # writes <out_dir>/<name>.pt
"""
import os
import sys
from collections import OrderedDict

import torch

BIG = 65536


def pack_state(state):
    out = OrderedDict()
    for k, v in state.items():
        v = v.detach().cpu()
        if v.dtype == torch.float32 and v.numel() >= BIG:
            v = v.to(torch.float16)
        out[k] = v
    return out


def unpack_state(state):
    return OrderedDict((k, v.to(torch.float32) if v.dtype == torch.float16 else v) for k, v in state.items())


def pack(path, named_states):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(OrderedDict((name, pack_state(sd)) for name, sd in named_states.items()), path)
    return os.path.getsize(path)


def unpack(path, out_dir=None):
    blob = torch.load(path, map_location="cpu")
    states = OrderedDict((name, unpack_state(sd)) for name, sd in blob.items())
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        for name, sd in states.items():
            torch.save(sd, os.path.join(out_dir, name + ".pt"))
    return states


def main(argv):
    if len(argv) >= 3 and argv[0] == "pack":
        named = OrderedDict()
        for item in argv[2:]:
            name, src = item.split("=", 1)
            named[name] = torch.load(src, map_location="cpu")
        print("%s: %.1f MiB" % (argv[1], pack(argv[1], named) / 2.0 ** 20))
        return 0
    if len(argv) == 3 and argv[0] == "unpack":
        for name in unpack(argv[1], argv[2]):
            print(os.path.join(argv[2], name + ".pt"))
        return 0
    sys.stderr.write(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
