#!/usr/bin/env python3
"""Round 6: train the codec for real with the product's training path (train.py) on one
MI355X, on procedural ERP images, inside gpurun's 20-minute calls.

    python tools/train_round6.py stage1 --minutes 15 [--from trained/r6/stage1.pack.pt]
        transforms + quantiser (train.py --base: CMPNetV2M), viewport MSE + SSIM loss
    python tools/train_round6.py stage2 --minutes 8 --from trained/r6/stage1.pack.pt
        entropy model on the frozen codes (train.py --init: CMPNetV2MF, gradient cut at the symbols),
        then export.py -> the codec's three files, and an evaluation THROUGH THE CODEC (PseudoEncoder ->
        file -> PseudoDecoder, the engine path) on held-out procedural frames: bpp, viewport PSNR / SSIM

Every stage returns its weights packed under gpurun_out/train_r6/ (tools/weights_pack.py: big tensors
as fp16, 64 MiB return channel); copy them to trained/r6/ (git-ignored, travels to the GPU box) for the
next call.  Seeds are fixed: the run is reproducible up to the float atomics of the backward kernels.
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch

import weights_pack

OUT = os.path.join(ROOT, "gpurun_out", "train_r6")
VALID_DIM, CHANNELS, NPART = 56, 192, 16


def common(args, work):
    return ["--procedural", str(args.images), "--height", str(args.height), "--width", str(args.width),
            "--batch-size", str(args.batch_size), "--test-batch-size", "2", "--acc-batch", "1",
            "--valid-dim", str(VALID_DIM), "--channels", str(args.channels), "--code-dim", str(args.channels),
            "--epochs", "100000", "--time-budget", str(args.minutes * 60.0), "--gamma", "1", "--beta", str(args.beta),
            "--clip", str(args.clip), "--mean", "0", "--viewport_size", str(args.viewport), "--workers", str(args.workers),
            "--base-dir", work, "--seed", str(args.seed), "--verbose", "--device", args.device]


def newest(save_dir, prex):
    """the checkpoint with the best test loss (`<prex>_best_0.pt`, ModuleSaver), else `<prex>_latest.pt`.  (Round 6's
    first stage-1 call returned the LATEST state and with it the loss spike of its last epochs: 31.8 dB at epoch 9,
    15.5 dB at epoch 10 -- the best one is what a training run hands on.)"""
    for s in ("best_0", "latest"):
        cand = os.path.join(save_dir, "%s_%s.pt" % (prex, s))
        if os.path.exists(cand):
            return cand
    raise SystemExit("no checkpoint under %s" % save_dir)


def unpack_to(pack_path, name, dst):
    states = weights_pack.unpack(pack_path)
    torch.save(states[name], dst)


def stage1(args):
    from pseudocylindrical_convolution_amd import train
    work = args.work
    save = os.path.join(work, "save_models")
    os.makedirs(save, exist_ok=True)
    prex = "base_opt_%d_%d_%d" % (args.channels, VALID_DIM, NPART)
    argv = ["--base", "--lr", str(args.lr)] + common(args, work)
    if getattr(args, "from_"):
        init = os.path.join(save, "stage1_init.pt")
        unpack_to(args.from_, "train_state", init)
        argv += ["--init-from", init]
    t0 = time.time()
    train.main(argv)
    ckpt = newest(save, prex)
    state = torch.load(ckpt, map_location="cpu")
    size = weights_pack.pack(os.path.join(OUT, "stage1.pack.pt"), {"train_state": state})
    shutil.copy(os.path.join(save, "%s_logs_0.txt" % prex), os.path.join(OUT, "stage1_log_%d.txt" % int(t0)))
    print("stage1: %s -> gpurun_out/train_r6/stage1.pack.pt (%.1f MiB) in %.0f s" % (os.path.basename(ckpt), size / 2.0 ** 20,
                                                                                   time.time() - t0))


def evaluate(codec_dir, prex, args, frames=4):
    """bpp / viewport PSNR / SSIM of held-out procedural frames through the codec's real path"""
    import tempfile
    from pseudocylindrical_convolution_amd import pseudo_codec as PC
    from pseudocylindrical_convolution_amd.SphereDataset import ProceduralSphereDataSet
    dev = torch.device("cuda:0")
    enc = PC.PseudoEncoder(VALID_DIM, 0).to(dev).eval()
    dec = PC.PseudoDecoder(VALID_DIM, 0).to(dev).eval()
    PC.load_models(enc, "%s/%s_encoder.pt" % (codec_dir, prex), "%s/%s_ent.pt" % (codec_dir, prex), dev)
    PC.load_models(dec, "%s/%s_decoder.pt" % (codec_dir, prex), "%s/%s_ent.pt" % (codec_dir, prex), dev)
    data = ProceduralSphereDataSet(frames, args.height, args.width, seed=args.seed * 2 + 99)
    metrics = PC.ViewportMetrics(0)
    tmp = tempfile.mkdtemp()
    rows = []
    for i in range(frames):
        x = data[i].unsqueeze(0).to(dev)
        path = os.path.join(tmp, "f%d.bin" % i)
        enc(x, path)
        rec = dec(path, args.height, args.width)
        psnr, ssim = metrics(x, rec)
        erp_mse = torch.mean((x - rec) ** 2).item()
        rows.append({"bpp": os.path.getsize(path) * 8.0 / args.height / args.width, "viewport_psnr_db": float(psnr),
                     "viewport_ssim": float(ssim), "erp_psnr_db": -10.0 * torch.log10(torch.tensor(erp_mse)).item()})
    mean = {k: sum(r[k] for r in rows) / len(rows) for k in rows[0]}
    return {"frames": rows, "mean": mean}


def stage2(args):
    from pseudocylindrical_convolution_amd import export, train
    work = args.work
    save = os.path.join(work, "save_models")
    os.makedirs(save, exist_ok=True)
    # train.py --init reads the stage-1 transforms from save_models/base_opt_..._best_0.pt
    base = os.path.join(save, "base_opt_%d_%d_%d_best_0.pt" % (args.channels, VALID_DIM, NPART))
    unpack_to(args.from_, "train_state", base)
    prex = "ent_opt_%d_%d_%d_init" % (args.channels, VALID_DIM, NPART)
    t0 = time.time()
    train.main(["--init", "--lr", str(args.lr), "--alpha", "1"] + common(args, work))
    ckpt = newest(save, prex)
    # export.py masks the 5x5 weights with the backend's MaskConstrainOp: on the GPU box the state must live on the GPU
    # (call 2 of this round loaded it on the CPU and lost a finished stage 2 to "expected a GPU tensor")
    state = torch.load(ckpt, map_location="cuda:0" if args.device == "cuda" else "cpu")
    codec_dir = os.path.join(work, "codec")
    cprex = "3_%d" % VALID_DIM
    export.export_codec(state, VALID_DIM, codec_dir, cprex)
    named = {"%s_%s" % (cprex, n): torch.load(os.path.join(codec_dir, "%s_%s.pt" % (cprex, n)), map_location="cpu")
             for n in ("encoder", "decoder", "ent")}
    size = weights_pack.pack(os.path.join(OUT, "codec_%s.pack.pt" % cprex), named)
    # evaluate what later calls will load: the packed (fp16-rounded) weights
    rounded = os.path.join(work, "codec_rounded")
    weights_pack.unpack(os.path.join(OUT, "codec_%s.pack.pt" % cprex), rounded)
    report = {"stage2_seconds": round(time.time() - t0, 1), "pack_mib": round(size / 2.0 ** 20, 1)}
    if args.device == "cuda":
        report["codec_eval_%dx%d" % (args.height, args.width)] = evaluate(rounded, cprex, args)
    with open(os.path.join(OUT, "stage2_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    shutil.copy(os.path.join(save, "%s_logs_0.txt" % prex), os.path.join(OUT, "stage2_log_%d.txt" % int(t0)))
    print("stage2:", json.dumps(report.get("codec_eval_%dx%d" % (args.height, args.width), {}).get("mean", {})))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage", choices=["stage1", "stage2", "eval"])
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--from", dest="from_", default=None)
    ap.add_argument("--images", type=int, default=1024, help="procedural images per epoch")
    ap.add_argument("--batch-size", type=int, default=2)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=CHANNELS)
    ap.add_argument("--viewport", type=int, default=171)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--beta", type=float, default=0.01, help="weight of the viewport (1 - SSIM) term")
    ap.add_argument("--clip", type=float, default=1.0)
    ap.add_argument("--workers", type=int, default=10)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--work", default="/tmp/train_r6")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if args.stage == "stage1":
        stage1(args)
    elif args.stage == "stage2":
        if not args.from_:
            raise SystemExit("stage2 needs --from <stage1.pack.pt>")
        stage2(args)
    else:
        codec_dir = os.path.join(args.work, "codec_rounded")
        weights_pack.unpack(args.from_, codec_dir)
        print(json.dumps(evaluate(codec_dir, "3_%d" % VALID_DIM, args)["mean"]))


if __name__ == "__main__":
    main()
