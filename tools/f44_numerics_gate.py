#!/usr/bin/env python3
"""Numerics gate for Winograd F(4x4, 3x3): no kernel, a torch emulation on the CPU.

The whole codec graph (analysis + synthesis transforms, the product's model code on the CPU oracle backend -- test
infrastructure, this is a probe) is run with its 3x3 stride-1 convolutions replaced by fp32 emulations of the
Winograd forms, transforms and channel sums in float32 like the kernels':

    ref64   every convolution in float64, rounded to fp32 once per layer       (the yardstick)
    chain   plain fp32 convolution                                              (what the oracle / direct kernel do)
    today   F(4x2) on the layers wino42.hip takes (cin % 24 == 0, cout % 64 == 0), F(2x2) on the other 3x3 s1 layers
    f44     as `today`, but the 192 -> 192 layers at the half-resolution scale (the 200 ms / step class) on F(4x4)

Reported per variant: analysis code max abs error vs ref64, quantiser symbols that differ from ref64's, and the
reconstruction max abs error vs ref64 when every variant synthesises ref64's symbols.  Gate (VERDICT): F(4x4) is worth
a kernel only if the reconstruction stays <= 5e-5 and the ties do not grow.

    python tools/f44_numerics_gate.py [--weights DIR] [--height 512 --width 1024] [--frames 2]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                    [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
G4 = torch.tensor([[1 / 4., 0, 0], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.], [1 / 24., 1 / 12., 1 / 6.],
                   [1 / 24., -1 / 12., 1 / 6.], [0, 0, 1]], dtype=torch.float32)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)
FORMS = {2: (BT2, G2, AT2), 4: (BT4, G4, AT4)}


def winograd(x, weight, bias, mv, mh):
    """F(mv x mh, 3x3) of a stride-1 valid convolution, everything in float32"""
    (btv, gv, atv), (bth, gh, ath) = FORMS[mv], FORMS[mh]
    tn, cin, h, w = x.shape
    ho, wo = h - 2, w - 2
    th, tw = -(-ho // mv), -(-wo // mh)
    xp = F.pad(x, (0, tw * mh + 2 - w, 0, th * mv + 2 - h))
    d = xp.unfold(2, mv + 2, mv).unfold(3, mh + 2, mh)                       # tn, cin, th, tw, mv+2, mh+2
    v = torch.einsum("ia,ncyxab,jb->ijncyx", btv, d, bth)                    # V = Bt d B
    u = torch.einsum("ia,ocab,jb->ijoc", gv, weight, gh)                     # U = G g Gt
    m = torch.einsum("ijoc,ijncyx->ijnoyx", u, v)                            # sum over input channels
    y = torch.einsum("pi,ijnoyx,qj->noypxq", atv, m, ath)                    # Y = At M A
    y = y.reshape(tn, weight.shape[0], th * mv, tw * mh)[:, :, :ho, :wo]
    return (y + bias.view(1, -1, 1, 1)).contiguous() if bias is not None else y.contiguous()


def make_conv(variant, half_width):
    def conv(owner, x, weight, bias, stride, slope=None, col_limit=None, npart=0, **kw):
        cout, cin, k, _ = weight.shape
        w3 = k == 3 and stride == 1 and cin >= 16
        if variant == "ref64":
            y = F.conv2d(x.double(), weight.double(), bias.double() if bias is not None else None, stride).float()
        elif variant == "chain" or not w3:
            y = F.conv2d(x, weight, bias, stride)
        else:
            takes42 = cin % 24 == 0 and cout % 64 == 0
            if variant == "f44" and cin == 192 and cout == 192 and x.shape[3] - 2 >= half_width:
                y = winograd(x, weight.detach(), bias.detach() if bias is not None else None, 4, 4)
            elif takes42:
                y = winograd(x, weight.detach(), bias.detach() if bias is not None else None, 4, 2)
            else:
                y = winograd(x, weight.detach(), bias.detach() if bias is not None else None, 2, 2)
        return F.prelu(y, slope) if slope is not None else y
    return conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=None, help="directory with 3_56_{encoder,decoder,ent}.pt (default: seeded random)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from pseudocylindrical_convolution_amd.PCONV_operator import backend
    from oracle import pconv_cpu as O, coder_cpu
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    from pseudocylindrical_convolution_amd.SphereDataset import procedural_erp
    backend.use(O, coder_cpu)
    O.set_detmath(True)
    torch.manual_seed(1234)
    enc, dec = PC.PseudoEncoder(56, 0).eval(), PC.PseudoDecoder(56, 0).eval()
    if args.weights and args.weights.endswith(".pack.pt"):
        # a packed stage-1 training state (tools/train_round6.py): the transforms and the quantiser are all this probe needs
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import weights_pack
        st = weights_pack.unpack(args.weights)["train_state"]
        enc.encoder.load_state_dict({k[len("encoder."):]: v for k, v in st.items() if k.startswith("encoder.")})
        dec.decoder.load_state_dict({k[len("decoder."):]: v for k, v in st.items() if k.startswith("decoder.")})
        enc.quant.load_state_dict({k[len("quant."):]: v for k, v in st.items() if k.startswith("quant.")}, strict=False)
        dec.quant.weight.data.copy_(enc.quant.weight.data)
    elif args.weights:
        PC.load_models(enc, args.weights + "/3_56_encoder.pt", args.weights + "/3_56_ent.pt", "cpu")
        PC.load_models(dec, args.weights + "/3_56_decoder.pt", args.weights + "/3_56_ent.pt", "cpu")
    else:
        dec.quant.weight.data.copy_(enc.quant.weight.data)
    original = O.tile_conv2d
    report = {"weights": args.weights or "seeded random", "size": "%dx%d" % (args.height, args.width), "frames": args.frames,
              "variants": {}}
    half = args.width // 2
    worst = {}
    try:
        for f in range(args.frames):
            x = procedural_erp(args.height, args.width, 515151 + f, 1.0 + f).unsqueeze(0)
            res = {}
            for variant in ("ref64", "chain", "today", "f44"):
                O.tile_conv2d = make_conv(variant, half)
                with torch.no_grad():
                    code = enc.encoder(enc.slice(x)).clone()
                    _, code_i = enc.quant(code)
                    sym = enc.ent.fill(enc.dtw(enc.ext(code_i))).clone()
                res[variant] = {"code": code, "sym": sym}
            ref = res["ref64"]
            for variant in ("ref64", "chain", "today", "f44"):
                O.tile_conv2d = make_conv(variant, half)
                with torch.no_grad():
                    res[variant]["rec"] = dec.reconstruct(ref["sym"]).clone()      # every variant synthesises ref64's symbols
            for variant in ("chain", "today", "f44"):
                r = res[variant]
                row = {"analysis_code_max_abs_err": (r["code"] - ref["code"]).abs().max().item(),
                       "quantiser_ties": int((r["sym"] != ref["sym"]).sum()),
                       "reconstruction_max_abs_err": (r["rec"] - ref["rec"]).abs().max().item(),
                       "reconstruction_rms_err": (r["rec"] - ref["rec"]).pow(2).mean().sqrt().item()}
                w = worst.setdefault(variant, dict(row))
                for k, v in row.items():
                    w[k] = max(w[k], v) if k != "quantiser_ties" else (w[k] + v if f else v)
            report["code_scale_max_abs"] = max(report.get("code_scale_max_abs", 0.0), ref["code"].abs().max().item())
            report["symbols_per_frame"] = ref["sym"].numel()
    finally:
        O.tile_conv2d = original
        backend.reset()
    report["variants"] = worst
    f44, today = worst["f44"], worst["today"]
    report["gate"] = {"reconstruction_le_5e-5": f44["reconstruction_max_abs_err"] <= 5e-5,
                      "ties_do_not_grow": f44["quantiser_ties"] <= today["quantiser_ties"]}
    report["verdict"] = "F(4x4) passes the numerics gate" if all(report["gate"].values()) else "F(4x4) fails the numerics gate"
    text = json.dumps(report, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as fo:
            fo.write(text + "\n")


if __name__ == "__main__":
    main()
