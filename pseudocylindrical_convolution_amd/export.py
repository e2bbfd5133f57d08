"""Training checkpoint -> the codec's three files (`{prex}_encoder.pt`, `{prex}_decoder.pt`,
`{prex}_ent.pt`; reference: pseudo_codec.py:223-269 loads them with strict=True).

The reference publishes converted files but not the converter.  The mapping follows from the two
model definitions (model_zoo_v2.py:214-334 vs pseudo_codec.py:27-117):
  * `encoder.*`, `quant.weight`, `quant.count`   -> the encoder file, names unchanged
  * `decoder.*`, `quant.weight`                  -> the decoder file
  * the three EntropySubNets, stacked in the order the codec's batched GMM table reads its planes
    (weights | scales | means, entropy_gmm_table_cuda.cu:147-149) -> `ent.net.<layer>...` with a
    leading batch axis of 3; PReLU slopes become the `relu` vectors; the 5x5 weights are stored
    with their causal mask applied (the training module masks them on every forward).
The softmax / ReLU+1e-6 that end the weight / scale sub-nets live inside the codec's table kernel."""
import os
from collections import OrderedDict

import torch

from .PCONV_operator import backend

SUBNETS = ("weight_net", "delta_net", "mean_net")      # batch index 0, 1, 2 of the codec's entropy net


def _masked(weight, ngroup, hidden):
    w = weight.detach().clone().contiguous()
    dev = w.device
    gid = dev.index if dev.index is not None else 0
    backend.ops().MaskConstrainOp(6 if hidden else 5, ngroup, gid, False).forward(w)
    return w


def entropy_state(state, ngroup, prefix="ent."):
    """`ent.net.*` tensors of EntEncoder / EntDecoder from a CMPNetV2MF / CMPNetV2MFEntropy state dict"""
    out = OrderedDict()

    def stack(fmt):
        return torch.stack([state[prefix + s + "." + fmt] for s in SUBNETS], 0).contiguous()

    def conv(src, dst, hidden, act):
        w = torch.stack([_masked(state["%s%s.%s.conv.weight" % (prefix, s, src)], ngroup, hidden) for s in SUBNETS], 0)
        out[dst + ".weight"] = w.contiguous()
        out[dst + ".bias"] = stack(src + ".conv.bias")
        if act:
            out[dst + ".relu"] = stack(src + ".act.weight")

    conv("net.0", "ent.net.0.conv", False, True)
    for layer in range(1, 6):
        for c in ("conv1", "conv2"):
            conv("net.%d.%s" % (layer, c), "ent.net.%d.%s.conv" % (layer, c), True, True)
    conv("net.6", "ent.net.6.conv", True, False)
    return out


def codec_states(state, valid_dim):
    """(encoder file, decoder file, entropy file) state dicts"""
    enc = OrderedDict((k, v.detach().clone()) for k, v in state.items()
                      if k.startswith("encoder.") or k in ("quant.weight", "quant.count"))
    dec = OrderedDict((k, v.detach().clone()) for k, v in state.items() if k.startswith("decoder."))
    dec["quant.weight"] = state["quant.weight"].detach().clone()
    return enc, dec, entropy_state(state, valid_dim // 4)


def export_codec(model_or_state, valid_dim, out_dir, prex):
    """write `<out_dir>/<prex>_{encoder,decoder,ent}.pt`; returns the three paths"""
    state = model_or_state if isinstance(model_or_state, dict) else model_or_state.state_dict()
    if any(k.startswith("module.") for k in state):
        state = OrderedDict((k[len("module."):], v) for k, v in state.items())
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for name, sd in zip(("encoder", "decoder", "ent"), codec_states(state, valid_dim)):
        path = os.path.join(out_dir, "%s_%s.pt" % (prex, name))
        torch.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), path)
        paths.append(path)
    return paths


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="training checkpoint -> codec files")
    ap.add_argument("checkpoint")
    ap.add_argument("--valid-dim", type=int, required=True)
    ap.add_argument("--out-dir", default="./demo/mse")
    ap.add_argument("--prex", required=True, help="file prefix, e.g. 3_56")
    args = ap.parse_args(argv)
    state = torch.load(args.checkpoint, map_location="cuda:0" if torch.cuda.is_available() else "cpu")
    for p in export_codec(state, args.valid_dim, args.out_dir, args.prex):
        print(p)


if __name__ == "__main__":
    main()
